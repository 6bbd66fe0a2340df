"""How many CPUs the calling process may really use (host-side support for measurements and tests: no arithmetic of the hot path lives here).

A GPU box of the pool shows 256 logical CPUs to `os.cpu_count()` and an all-ones affinity mask, but its cgroup grants a
CPU-time quota of 16 (`/sys/fs/cgroup/cpu.max` = "1600000 100000").  torch then starts 128 intra-op threads that are
throttled to 16 CPUs' worth of time: a CPU fp64 step of ResNet-18 / 16 tile pairs of 64x64 measured 25.5 s with
the default 128 threads against 4.5 s with 16 (tools/host_probe.py, gpurun_out/r4_probe.log) -- that, not the
arithmetic, was most of the round-3 GPU suite's 960 s.  Every CPU-side timing or checker run sizes its thread pool with
`usable_cpus()`."""
import os


def usable_cpus() -> int:
    """min(affinity mask, cgroup v2 / v1 CPU quota), at least 1"""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(int(q) / int(p))))
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // p))
        except (OSError, ValueError):
            pass
    return max(1, n)


def set_torch_threads(reserve: int = 0) -> int:
    """size torch's intra-op pool to the usable CPUs (minus `reserve` kept for other processes); returns the count"""
    import torch

    n = max(1, usable_cpus() - reserve)
    if torch.get_num_threads() != n:
        torch.set_num_threads(n)
    return n
