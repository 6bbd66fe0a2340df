"""Host-side probe of a GPU box: how many CPUs the process may really use, and how the CPU oracle's step time moves with
the torch thread count (tests/oracle workers are sized from this)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
print("os.cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
for p in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us",
          "/sys/fs/cgroup/cpuset.cpus.effective", "/sys/fs/cgroup/memory.max"):
    try:
        print(p, open(p).read().strip())
    except OSError as e:
        print(p, "n/a")
import torch

print("torch threads default", torch.get_num_threads(), flush=True)
from helpers import LR, WEIGHTS, build_case, case_batch, load_golden
from oracle import msfwsi_oracle as orc

case = sys.argv[1] if len(sys.argv) > 1 else "r18_b16_s64_div"
vec, man = load_golden(case)
t0 = time.time()
model = build_case(man)
print("build", case, round(time.time() - t0, 2), flush=True)
sd0 = {k: v.detach().clone() for k, v in model.state_dict().items()}
batch = case_batch(man)
for dt in (torch.float64, torch.float32):
    for th in (4, 8, 16, 32, 64, 128):
        if th > os.cpu_count():
            continue
        torch.set_num_threads(th)
        sd = {k: (v.clone().to(dt) if v.is_floating_point() else v.clone()) for k, v in sd0.items()}
        (c1, c2), (t1, t2), idx = batch
        b = ((c1.to(dt), c2.to(dt)), (t1.to(dt), t2.to(dt)), idx)
        lr = orc.init_lr(LR, man["B"])
        t0 = time.time()
        orc.train_step(sd, b, orc.Adam(sd, [lr, lr, lr]), 4, 0.5, WEIGHTS)
        print(case, dt, "threads", th, "step s", round(time.time() - t0, 2), flush=True)
