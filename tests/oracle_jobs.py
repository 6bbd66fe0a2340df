"""CPU oracle work of the GPU suite, run in BACKGROUND worker processes so that it overlaps the GPU tests.

Test infrastructure only (like oracle/): nothing under msf_wsi_amd/ imports this.  Why it exists: the round-3 suite spent
~500 of its 960 s waiting for the fp64 / fp32 CPU oracle (tests/helpers.oracle_case and friends) while the GPU idled, and
the driver's 900 s limit cut it off.  The arithmetic is unchanged -- the same oracle functions on the
same seeded inputs -- only WHERE and WHEN they run:

  * `start(names)` (tests/conftest.py, once per session that selected GPU tests) spawns a few CPU-only worker processes
    (spawn context, CUDA/HIP devices hidden: they never touch the card, so they do not count against the box's
    GPU-process guard) and queues the named jobs, largest first;
  * `get(name)` returns a job's result: waits for the worker, or -- when no pool was started (CPU suite, a single test run
    by hand) -- computes it inline in the calling process;
  * results travel as a torch.save file under /dev/shm (or $TMPDIR) that the parent maps (`mmap=True`) and unlinks: the
    13 GB fp64 gradient sets of the ResNet-50-derived case are not pickled through a pipe.

Every job is a top-level function of this module, so that a spawned worker can import it.
"""
import atexit
import os
import shutil
import sys
import tempfile
import time
from concurrent.futures import ProcessPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)

_POOL = None
_FUTS = {}
_DIR = None
_CACHE = {}
_T0 = time.time()


# --------------------------------------------------------------------------------------------------
# the jobs
# --------------------------------------------------------------------------------------------------
def _setup_worker(threads):
    os.environ["CUDA_VISIBLE_DEVICES"] = ""
    os.environ["HIP_VISIBLE_DEVICES"] = ""
    for p in (ROOT, HERE):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch

    torch.set_num_threads(threads)


def job_oracle_case(case):
    """helpers.oracle_case's arithmetic: one fp64 and one fp32 oracle step (forward, loss, backward, Adam) of a golden
    case.  The fp32 run only feeds the per-tensor fp32<->fp64 spread (`box_grad`, `box_step`) and its loss: its
    gradients / weights are not shipped back."""
    import numpy as np
    import torch

    from helpers import LR, WEIGHTS, build_case, case_batch, load_golden, rel
    from oracle import msfwsi_oracle as orc

    vec, man = load_golden(case)
    B, size = man["B"], man["size"]
    model = build_case(man)
    sd0 = {k: v.detach().clone() for k, v in model.state_dict().items()}
    names = [n for n, _ in model.named_parameters()]
    del model
    batch = case_batch(man)
    lr = orc.init_lr(LR, B)
    out = {"B": B, "size": size, "lr": lr, "names": names, "sd0": sd0}
    keep = {}
    for tag, dt in (("64", torch.float64), ("32", torch.float32)):
        sd = {k: (v.clone().to(dt) if v.is_floating_point() else v.clone()) for k, v in sd0.items()}
        (c1, c2), (t1, t2), idx = batch
        b = ((c1.to(dt), c2.to(dt)), (t1.to(dt), t2.to(dt)), idx)
        loss, terms, outs, grads = orc.train_step(sd, b, orc.Adam(sd, [lr, lr, lr]), 4, 0.5, WEIGHTS)
        out["loss" + tag] = float(loss)
        out["terms" + tag] = torch.stack([t for row in terms for t in row]).detach()
        if tag == "64":
            out["outs64"] = tuple(tuple(tuple(t.detach() for t in tup) for tup in grp) for grp in outs)
            out["grads64"] = {k: v.detach() for k, v in grads.items()}
        out["sd" + tag] = sd
        keep[tag] = (grads, sd)
    out["box_grad"] = np.array([rel(keep["32"][0][n], keep["64"][0][n]) for n in names])
    out["box_step"] = np.array([rel(keep["32"][1][n], keep["64"][1][n]) for n in names])
    return out


def job_steps(case, steps, dtype_name, loss="cosine", temperature=0.2, want_grads=False):
    """`steps` consecutive oracle steps (with Adam) of a golden case's seeded model and N(0,1) batch in one precision:
    {losses, sd (weights after the last step), grads (of the last step, optional)} -- tests/test_train_gpu.py"""
    import torch

    from helpers import LR, WEIGHTS, build_case, load_golden
    from oracle import msfwsi_oracle as orc

    dt = {"fp64": torch.float64, "fp32": torch.float32}[dtype_name]
    vec, man = load_golden(case)
    B, size = man["B"], man["size"]
    model = build_case(man)
    sd = {k: (v.detach().clone().to(dt) if v.is_floating_point() else v.detach().clone())
          for k, v in model.state_dict().items()}
    del model
    (c1, c2), (t1, t2), idx = orc.synthetic_batch(B, size, 16, man["data_seed"])
    b = ((c1.to(dt), c2.to(dt)), (t1.to(dt), t2.to(dt)), idx)
    lr = orc.init_lr(LR, B)
    if loss == "cosine":
        opt, kw = orc.Adam(sd, [lr, lr, lr]), {}
    else:
        opt = type("NoOpt", (), {"step": lambda self, *a, **k: None})()
        kw = {"loss_fn": lambda o, w: orc.infonce_terms(o, w, temperature=temperature)}
    losses, grads = [], None
    for _ in range(steps):
        l_, _, _, grads = orc.train_step(sd, b, opt, 4, 0.5, WEIGHTS, **kw)
        losses.append(float(l_))
    out = {"losses": losses, "sd": sd}
    if want_grads:
        out["grads"] = {k: v.detach() for k, v in grads.items()}
    return out


def job_curve_oracle(case, dtype_name):
    """the ORACLE under torch.autocast("cpu", 16-bit) on THIS machine's CPU over the curve fixture's steps: one more
    reference-under-autocast sample of the chaotic 30-step trajectory (16-bit CPU kernels differ by CPU generation) --
    tests/test_lowp_parity_gpu.py::test_loss_curve_tracks_reference"""
    import numpy as np
    import torch

    from helpers import LR, build_case, load_golden
    from oracle import msfwsi_oracle as orc

    vec, man = load_golden(case)
    B, size = man["B"], man["size"]
    dt = {"bf16": torch.bfloat16, "fp16": torch.float16}[dtype_name]
    steps = len(vec[f"loss_{dtype_name}"])
    osd = {k: v.detach().clone() for k, v in build_case(man).state_dict().items()}
    oopt = orc.Adam(osd, [orc.init_lr(LR, B)] * 3)
    here = []
    for t in range(steps):
        l_, _, _, _ = orc.train_step(osd, orc.diverse_batch(B, size, 16, man["curve_seed0"] + t), oopt, autocast_dtype=dt)
        here.append(float(l_))
    return {"losses": np.array(here)}


JOBS = {
    "oracle_case": job_oracle_case,
    "steps": job_steps,
    "curve_oracle": job_curve_oracle,
}


def _run(name, key, path, threads):
    """worker side: compute, save, return the path"""
    _setup_worker(threads)
    import torch

    t0 = time.time()
    out = JOBS[name](*key)
    torch.save(out, path)
    return path, time.time() - t0


# --------------------------------------------------------------------------------------------------
# the pool
# --------------------------------------------------------------------------------------------------
def usable_cpus():
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    from msf_wsi_amd.hostcpu import usable_cpus as f

    return f()


def start(requests, workers=None, threads=None):
    """spawn the workers and queue `requests` = [(job name, args tuple), ...] in the given order (put the long ones
    first).  Safe to call once per process; a second call only queues what is new."""
    global _POOL, _DIR
    import multiprocessing as mp

    # a GPU box grants 16 CPUs' worth of time (msf_wsi_amd/hostcpu.py).  The oracle's small-batch steps scale poorly with
    # threads (fp64 ResNet-18 step: 6.6 s on 4 threads, 4.5 s on 16), so several jobs side by side on 4 threads each
    # finish far sooner than one after the other on all of them; the test process keeps the rest for itself
    cpus = usable_cpus()
    if workers is None:
        workers = max(1, min(3, cpus // 4 - 1)) if cpus >= 8 else 1
    if threads is None:
        threads = max(1, min(6, (cpus - 4) // workers)) if cpus >= 8 else max(1, cpus // 2)
    if _POOL is None:
        base = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else None
        _DIR = tempfile.mkdtemp(prefix="msfwsi_oracle_", dir=base)
        atexit.register(stop)
        _POOL = ProcessPoolExecutor(max_workers=workers, mp_context=mp.get_context("spawn"))
        import torch

        main_threads = max(2, cpus - workers * threads + 2)
        torch.set_num_threads(main_threads)
        print(f"[oracle_jobs] {workers} CPU workers x {threads} threads + {main_threads} in the test process "
              f"({cpus} usable CPUs), results under {_DIR}", file=sys.stderr)
    for name, key in requests:
        k = (name, tuple(key))
        if k not in _FUTS and k not in _CACHE:
            path = os.path.join(_DIR, f"{len(_FUTS):03d}_{name}.pt")
            _FUTS[k] = _POOL.submit(_run, name, tuple(key), path, threads)


def stop():
    global _POOL, _DIR
    if _POOL is not None:
        _POOL.shutdown(wait=False, cancel_futures=True)
        _POOL = None
    if _DIR is not None:
        shutil.rmtree(_DIR, ignore_errors=True)
        _DIR = None


def get(name, *key):
    """the result of job `name(*key)`: from the session cache, from a background worker (waits for it), or computed
    inline when no worker was asked for it"""
    import torch

    k = (name, tuple(key))
    if k in _CACHE:
        return _CACHE[k]
    fut = _FUTS.pop(k, None)
    out = None
    if fut is not None:
        t0 = time.time()
        try:
            path, took = fut.result()
            out = torch.load(path, mmap=True, weights_only=False)
            os.unlink(path)  # the mapping keeps the pages
            print(f"[oracle_jobs] {name}{key}: worker took {took:.1f} s, this test waited {time.time() - t0:.1f} s "
                  f"(t+{time.time() - _T0:.0f} s)", file=sys.stderr)
        except Exception as e:  # a worker died (out of memory, killed): the same arithmetic inline, never a skipped check
            print(f"[oracle_jobs] {name}{key}: worker failed ({type(e).__name__}: {e}); computing inline", file=sys.stderr)
            out = None
    if out is None:
        for p in (ROOT, HERE):
            if p not in sys.path:
                sys.path.insert(0, p)
        out = JOBS[name](*key)
    _CACHE[k] = out
    if _POOL is not None and all(f.done() for f in _FUTS.values()):
        torch.set_num_threads(usable_cpus())  # the workers are idle from here on: the test process takes their CPUs
    return out
