#!/usr/bin/env python
"""Headline benchmark: tile-pairs/s through one MSF-WSI pre-train step on MI355X.

    python bench.py [--gpus N --steps K --warmup W]          (N>1: launched by torch.distributed.run)

Workload = BASELINE.json configs[1]: ResNet-50 dual-stream, bf16, 256 synthetic tile pairs per GPU
(1 tile pair = 1 context + 16 target tiles x 2 views = 34 image passes of 224x224x3), weak scaling.
A "step" = forward + 12-term cosine loss + backward + gradient averaging + Adam, inputs resident in HBM.
Besides the contract line it reports
  roofline     : the dominant dense kernel family, timed live with HIP events on the launch stream
  cpu_baseline : the CPU oracle (oracle/, a port of the reference step) on a bounded sample, rank 0, N=1.
"""
import argparse
import json
import os
import sys
import time

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC (RCCL needs it on this driver); before any GPU call

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_TFLOPS = {"bf16": 2500.0, "fp16": 2500.0, "fp32": 157.3}   # dense MFMA peaks, MI355X_MICROARCH.md
PEAK_HBM_GBS = 8000.0
FLOP_PER_PAIR = {"resnet18": 363.37e9, "resnet50": 848.8e9}  # SURVEY.md 8(d), fwd+bwd per tile pair
# algorithmic minimum HBM bytes -- activations per tile pair (2-byte storage; x2 for fp32) and the
# batch-independent Adam pass (28 B per parameter) per step
ACT_BYTES_PER_PAIR_16BIT = {"resnet18": 1.01e9, "resnet50": 4.53e9}
ADAM_BYTES_PER_STEP = {"resnet18": 3.46e9, "resnet50": 46.6e9}


def _pmc_summary():
    """(table, file name, stale): the newest committed counter summary (profiles/r*_pmc.json: separate rocprofv3 --pmc passes
    of this same command, tools/profile_step.sh + tools/pmc_summary.py) -- but ONLY when it was taken with the library this
    process runs: the summary carries the build id of the binary it profiled (sha256 of the kernel sources,
    msfwsi_build_id), and a summary of another build is not replayed beside a fresh time (`stale` names the file then)."""
    import glob

    from msf_wsi_amd import _lib

    paths = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc.json")), key=lambda p: (os.path.getmtime(p), p))
    if not paths:
        return None, None, None
    mine = _lib.load().msfwsi_build_id().decode()
    newest = os.path.basename(paths[-1])
    for path in reversed(paths):
        try:
            with open(path) as f:
                tab = json.load(f)
        except (OSError, ValueError):
            continue
        if tab.get("build_id") == mine:
            return tab, os.path.basename(path), None
    return None, None, newest


def pmc_entry(symbol):
    """counter-derived figures of the kernel whose mangled name contains `symbol`: L2-miss (fabric) bytes per launch and MFMA
    utilisation, from the committed counter summary of THIS build ({} plus `stale_profile` when there is none)"""
    tab, name, stale = _pmc_summary()
    if tab is None:
        return {"stale_profile": stale} if stale else {}
    try:
        kern = tab.get("kernels", {})
        hit = sorted(((len(k), v) for k, v in kern.items() if k.strip() and (k in symbol or symbol in k)),
                     key=lambda kv: -kv[0])  # the longest (most specific) matching name
        return dict(hit[0][1], source=name) if hit else {}
    except (ValueError, KeyError):
        return {}


def pmc_step():
    """(whole-step counter traffic in bytes, summary file, stale file): L2-miss bytes per step of this build's summary"""
    tab, name, stale = _pmc_summary()
    if tab is None:
        return None, None, stale
    try:
        return float(tab["hbm_bytes_per_step"]), name, None
    except (ValueError, KeyError):
        return None, None, name


def _build_id():
    from msf_wsi_amd import _lib

    return _lib.load().msfwsi_build_id().decode()


def hub_stub():
    """random-init 'pretrained' weights: the GPU box has no network (data/weights are synthetic)"""
    from msf_wsi_amd.models import resnet as R

    def fake(url, progress=True, **kw):
        arch = [k for k, v in R.model_urls.items() if v == url][0]
        return R.__dict__[arch](pretrained=False).state_dict()

    torch.hub.load_state_dict_from_url = fake


def build(arch, device, rank=0):
    """rank r seeds its replica with 3407 + 1000 r: like the reference's mp.spawn workers (tools/ssl_train.py:46-48,68 seed
    the parent only) the ranks do NOT start from equal weights -- PretrainStep's rank-0 broadcast (DDP's, :170) makes them
    equal, and a run that lost that broadcast would show it in the loss"""
    from msf_wsi_amd.models import resnet as R
    from msf_wsi_amd.models.backbone import MSFWSI

    hub_stub()
    torch.manual_seed(3407 + 1000 * rank)
    with torch.device(device):
        model = MSFWSI(R.__dict__[arch], 4)
    return model.train()


def _cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for ln in f:
                if ln.startswith("model name"):
                    return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def _oracle_steps(arch, B, size, n_timed, budget_s):
    """1 warm-up + up to n_timed timed steps of the oracle's train_step (fp32) on B tile pairs; returns (pairs/s, n)"""
    from oracle import msfwsi_oracle as orc
    from msf_wsi_amd.models import resnet as R
    from msf_wsi_amd.models.backbone import MSFWSI

    hub_stub()
    torch.manual_seed(3407)
    model = MSFWSI(R.__dict__[arch], 4)
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    del model
    batch = orc.synthetic_batch(B, size, 16, 0)
    lr = orc.init_lr(1e-3, B)
    opt = orc.Adam(sd, [lr, lr, lr])
    orc.train_step(sd, batch, opt)  # warm-up (allocator, oneDNN primitive caches)
    n, t0 = 0, time.time()
    while n < n_timed:
        orc.train_step(sd, batch, opt)
        n += 1
        if time.time() - t0 > budget_s:
            break
    return B * n / (time.time() - t0), n


def cpu_baseline(arch, size, budget_s, threads):
    """SURVEY.md 8(d) / BASELINE.md 3: the oracle (a port of the reference step, verified against the real reference)
    on BASELINE config 1 EXACTLY -- ResNet-18 dual-stream, 8 tile pairs of 224x224, fp32 -- 1 warm-up + 3 timed steps
    on the CPUs the process may use (`cores` = the torch threads actually used = min(affinity, cgroup CPU quota): a GPU
    box of this pool shows 256 logical CPUs but grants 16; the round-3 figure ran 128 throttled threads and was 3-5x
    too slow).  `extra` carries one step of the bench's own architecture at 2 tile pairs, labelled."""
    torch.set_num_threads(threads)
    val, n = _oracle_steps("resnet18", 8, 224, 3, max(budget_s, 60.0))
    out = {"value": round(val, 4), "unit": "tile-pairs/s", "cores": threads, "kind": "port",
           "sample": f"{n} timed step(s) after 1 warm-up of the oracle (port of the reference step) on BASELINE config 1: "
                     f"resnet18, 8 tile pairs of 224x224, fp32",
           "cpu_model": _cpu_model(), "torch": torch.__version__,
           "blas": "mkl" if torch.backends.mkl.is_available() else "other"}
    if arch != "resnet18":
        v2, n2 = _oracle_steps(arch, 2, size, 1, budget_s)
        out["extra"] = {"value": round(v2, 4), "unit": "tile-pairs/s",
                        "sample": f"{n2} step of the oracle on 2 tile pairs of {arch} ({size}x{size}, fp32): the bench "
                                  f"architecture at a CPU-sized batch, not the metric's configuration"}
    return out


def build_roofline(timer, args, dt, timed_from, step_bytes, step_flop, use_pmc=True, nst=None, ev_seconds=None):
    # nst / ev_seconds: number and wall time of the event-bracketed steps when they are not the last (steps - timed_from)
    # steps of the timed region (multi-stream runs: one extra single-stream step after it, see main)
    """the `roofline` object of the bench line from the HIP-event timings of the last (steps - timed_from) steps:
    the dominant kernel symbol by time, its algorithmic FLOP/s and GB/s against the gfx950 peaks, the replayed counter
    figures of the committed rocprofv3 --pmc summary (config 2 only), whole-step fractions (where SURVEY 8(d) gives the
    step's algorithmic bytes / FLOPs), the top kernels and the per-family sums"""
    fam = timer.summary()
    summ = timer.summary(by_symbol=True)
    dom = max(summ, key=lambda k: summ[k]["seconds"])  # the kernel (template instance) with the most time
    s = summ[dom]
    tf = s["flops"] / s["seconds"] / 1e12
    gbs = s["bytes"] / s["seconds"] / 1e9
    frac_m, frac_h = tf / PEAK_TFLOPS[args.dtype], gbs / PEAK_HBM_GBS
    bound = "mfma" if frac_m >= frac_h else "hbm"
    pmc = pmc_entry(dom) if use_pmc else {}
    nst = (args.steps - timed_from) if nst is None else nst
    # the counters are per rocprofv3 launch; one timer entry can issue several launches (the stride-2 3x3 input
    # gradient is four): compare per STEP, and quote `traffic` per timer entry like `achieved`
    pmc_per_step = (pmc["hbm_bytes_per_launch"] * pmc["launches_in_pass"] / 2.0
                    if pmc.get("hbm_bytes_per_launch") and pmc.get("launches_in_pass") else None)
    step_s = dt / args.steps
    step_traffic, step_src, step_stale = pmc_step() if use_pmc else (None, None, None)
    roof = {
        "kernel": dom, "family": s["family"], "bound": bound,
        "achieved": round(tf if bound == "mfma" else gbs, 2),
        "peak": PEAK_TFLOPS[args.dtype] if bound == "mfma" else PEAK_HBM_GBS,
        "unit": "TFLOP/s" if bound == "mfma" else "GB/s",
        "frac": round(max(frac_m, frac_h), 4),
        # replayed from the committed rocprofv3 --pmc summary of this same command (NOT measured in this run)
        "traffic": round(pmc_per_step * nst / s["launches"]) if pmc_per_step else None,
        "traffic_per_step": round(pmc_per_step) if pmc_per_step else None,
        "algorithmic_bytes_per_step": round(s["bytes"] / nst),
        "mfma_util": round(pmc["mfma_util"], 4) if pmc.get("mfma_util") is not None else None,
        "replayed_from": pmc.get("source"),
        # a committed counter summary exists, but of ANOTHER build of the kernels (its build id differs from this library's):
        # it is named, not replayed
        "stale_profile": pmc.get("stale_profile") or step_stale,
        "build_id": _build_id(),
        "whole_step": None if not (step_bytes and step_flop) else {
            "algorithmic_GB": round(step_bytes / 1e9, 1), "algorithmic_TFLOP": round(step_flop / 1e12, 1),
            "frac_hbm": round(step_bytes / step_s / 1e9 / PEAK_HBM_GBS, 4),
            "frac_mfma": round(step_flop / step_s / 1e12 / PEAK_TFLOPS[args.dtype], 4),
            "traffic_GB": round(step_traffic / 1e9, 1) if step_traffic else None,
            "traffic_ratio": round(step_traffic / step_bytes, 3) if step_traffic and step_bytes else None,
            "traffic_replayed_from": step_src},
        "launches": s["launches"], "avg_launch_ms": round(1e3 * s["seconds"] / s["launches"], 4),
        "algorithmic_bytes_per_launch": round(s["bytes"] / s["launches"]),
        "alt": {"TFLOP/s": round(tf, 2), "frac_mfma": round(frac_m, 4), "GB/s": round(gbs, 1),
                "frac_hbm": round(frac_h, 4)},
        "kernels": {k: {"launches": v["launches"], "ms": round(1e3 * v["seconds"], 2),
                        "TFLOP/s": round(v["flops"] / v["seconds"] / 1e12, 2),
                        "GB/s": round(v["bytes"] / v["seconds"] / 1e9, 1)}
                    for k, v in sorted(summ.items(), key=lambda kv: -kv[1]["seconds"])[:8]},
        "families": {k: {"launches": v["launches"], "ms": round(1e3 * v["seconds"], 2),
                         "TFLOP/s": round(v["flops"] / v["seconds"] / 1e12, 2),
                         "GB/s": round(v["bytes"] / v["seconds"] / 1e9, 1)} for k, v in fam.items()},
        "event_timed_steps": nst,
        "timed_fraction_of_step": round(sum(v["seconds"] for v in fam.values())
                                        / (ev_seconds if ev_seconds else dt * nst / args.steps), 3),
    }
    return roof


def stock_gpu_baseline(arch, size, B, steps=5, warmups=3):
    """A same-box yardstick beside cpu_baseline: the ORACLE (plain functional PyTorch: F.conv2d / F.batch_norm / F.linear,
    torch autograd, its own Adam) run on cuda:0 by stock PyTorch-ROCm eager -- MIOpen / hipBLASLt kernels under
    torch.autocast("cuda", bfloat16) -- at the largest batch that fits eager autograd's activation memory (stated).
    Protocol: MIOpen in its DEFAULT find mode with workspace (round 3 forced MIOPEN_FIND_MODE=FAST
    and timed 2 steps after 1 warm-up: that measured MIOpen's zero-workspace fallback kernels, 5 TFLOP/s), 3 warm-up
    steps (kernel search, allocator), 5 timed steps.  OPT-IN (--stock-batch N): the kernel search of the warm-up takes
    minutes.  Not the target and not the product: the product path never touches it."""
    import gc

    from oracle import msfwsi_oracle as orc
    from msf_wsi_amd.models import resnet as R
    from msf_wsi_amd.models.backbone import MSFWSI

    dev = torch.device("cuda", torch.cuda.current_device())
    hub_stub()
    torch.manual_seed(3407)
    model = MSFWSI(R.__dict__[arch], 4)
    sd = {k: v.detach().to(dev) for k, v in model.state_dict().items()}
    del model
    gc.collect()
    (c1, c2), (t1, t2), idx = orc.synthetic_batch(B, size, 16, 0)
    batch = ((c1.to(dev), c2.to(dev)), (t1.to(dev), t2.to(dev)), [i.to(dev) for i in idx])
    lr = orc.init_lr(1e-3, B)
    opt = orc.Adam(sd, [lr, lr, lr])
    torch.cuda.reset_peak_memory_stats()
    for _ in range(warmups):  # MIOpen kernel search (default find mode), hipBLASLt heuristics, allocator
        orc.train_step(sd, batch, opt, autocast_dtype=torch.bfloat16)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        orc.train_step(sd, batch, opt, autocast_dtype=torch.bfloat16)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return {"value": round(B * steps / dt, 3), "unit": "tile-pairs/s", "ms_per_step": round(1e3 * dt / steps, 1),
            "batch": B, "kind": "oracle on cuda (PyTorch-ROCm eager: MIOpen / hipBLASLt, bf16 autocast)",
            "miopen_find_mode": os.environ.get("MIOPEN_FIND_MODE", "default"),
            "sample": f"{steps} timed step(s) after {warmups} warm-ups, {arch}, {B} tile pairs of {size}x{size} (eager autograd "
                      f"keeps every activation: the bench batch of the product does not fit)",
            "peak_mem_GiB": round(torch.cuda.max_memory_allocated() / 2 ** 30, 1), "torch": torch.__version__}


def finetune_cpu_baseline(arch, classes, size, threads, budget_s):
    """the oracle's fine-tune step (oracle/hooknet_oracle.py: HookNet forward + Dice + backward, a port of the reference
    loop tools/ssl_finetune.py:431-458; smp's arithmetic restated, parity unpinned) on the host cores, fp32, 8 tile pairs"""
    from msf_wsi_amd.models.hooknet import HookNet
    from oracle import hooknet_oracle as ho
    from oracle import msfwsi_oracle as orc

    torch.set_num_threads(threads)
    torch.manual_seed(3407)
    model = HookNet(encoder_name=arch, encoder_weights=None, classes=classes + 1)
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    del model
    B = 8
    g = torch.Generator().manual_seed(0)
    x1, x2 = torch.randn(B, 3, size, size, generator=g), torch.randn(B, 3, size, size, generator=g)
    m1 = torch.randint(0, classes + 1, (B, size, size), generator=g)
    m2 = torch.randint(0, classes + 1, (B, size, size), generator=g)
    params = [v for k, v in sd.items() if orc.is_param(k)]
    for v in params:
        v.requires_grad_(True)
    opt = torch.optim.Adam(params, 1e-3)

    def step():
        loss, _ = ho.finetune_loss(sd, x1, x2, m1, m2, list(range(1, classes + 1)), 1.0)
        opt.zero_grad()
        loss.backward()
        opt.step()

    step()
    n, t0 = 0, time.time()
    while n < 3:
        step()
        n += 1
        if time.time() - t0 > budget_s:
            break
    return {"value": round(B * n / (time.time() - t0), 4), "unit": "tile-pairs/s", "cores": threads, "kind": "port",
            "sample": f"{n} timed step(s) after 1 warm-up of the oracle's fine-tune step (HookNet {arch}, Dice, torch Adam) on "
                      f"8 tile pairs of {size}x{size}, fp32", "cpu_model": _cpu_model(), "torch": torch.__version__}


def run_finetune(args, world, rank, dev, dtype):
    """BASELINE config 5: one fine-tune step = HookNet forward (context U-Net, hook, target U-Net), Dice loss on both
    logit maps, confusion counts of the target prediction, backward, gradient averaging, GradScaler + Adam
    (tools/ssl_finetune.py:422-462) on synthetic 256x256 tile pairs resident in HBM"""
    import torch.distributed as dist

    from msf_wsi_amd import kernels as kn
    from msf_wsi_amd.finetune import FinetuneStep
    from msf_wsi_amd.models.hooknet import HookNet

    torch.manual_seed(3407)
    with torch.device(dev):
        model = HookNet(encoder_name=args.arch, encoder_weights=None, classes=args.classes + 1).train()
    ts = FinetuneStep(model, lr=1e-3, batch_size=args.batch * world, dtype=dtype)
    g = torch.Generator(device=dev).manual_seed(rank)
    B, S = args.batch, args.size
    images = (torch.randn(B, 3, S, S, generator=g, device=dev), torch.randn(B, 3, S, S, generator=g, device=dev))
    masks = (torch.randint(0, args.classes + 1, (B, S, S), generator=g, device=dev),
             torch.randint(0, args.classes + 1, (B, S, S), generator=g, device=dev))

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        ts.step(images, masks)
    timer = None if args.no_kernel_timer else kn.KernelTimer(streams=False)
    sync()
    timed_from = max(0, args.steps - 2)
    t0 = time.perf_counter()
    for i in range(args.steps):
        kn.TIMER = timer if i >= timed_from else None
        loss, _ = ts.step(images, masks)
    sync()
    dt = time.perf_counter() - t0
    kn.TIMER = None
    tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())
    if rank != 0:
        return
    pairs = B * world * args.steps
    out = {"metric": "fine-tune tile-pairs/sec per step (HookNet, Dice, Adam)", "value": round(pairs / dt, 3),
           "unit": "tile-pairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
           "ms_per_step": round(1e3 * dt / args.steps, 3), "higher_is_better": True, "scaling": "weak",
           "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
           "config": {"workload": f"BASELINE config 5: HookNet ({args.arch} encoders, trainable) fine-tune step, {B} tile "
                                  f"pairs/GPU of {S}x{S}x3 (context + target), {args.classes} classes + background, "
                                  f"{args.dtype}, Dice + Adam + GradScaler, DP over {world} GPU(s)",
                      "arch": args.arch, "batch_per_gpu": B, "image_size": S, "parallelism": f"dp{world}",
                      "loss": float(loss), "peak_mem_GiB": round(torch.cuda.max_memory_allocated() / 2 ** 30, 1),
                      "note": "the reference supports 256x256 inputs only (hooknet.py:29-32) and trains the encoders "
                              "(ssl_finetune.py:289): BASELINE.json's '512x512 / frozen encoder' wording does not "
                              "match the code (SURVEY D5)"}}
    if timer is not None:
        out["roofline"] = build_roofline(timer, args, dt, timed_from, 0, 0, use_pmc=False)
    if world == 1 and not args.no_cpu_baseline:
        del ts, model
        torch.cuda.empty_cache()
        from msf_wsi_amd.hostcpu import usable_cpus

        threads = usable_cpus()
        out["cpu_baseline"] = finetune_cpu_baseline(args.arch, args.classes, S, threads, args.cpu_budget)
    print(json.dumps(out), flush=True)


def dmabuf_ipc_env():
    """the host driver only supports dmabuf IPC: without HSA_ENABLE_IPC_MODE_LEGACY=0 RCCL fails at
    hipIpcGetMemHandle.  Set before the first GPU call of EVERY rank, whoever launched it."""
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")


def launch_ranks(n):
    """one child process per GPU on this node over 127.0.0.1 (the container hostname may not resolve)"""
    import socket
    import subprocess

    port = os.environ.get("MASTER_PORT")
    if not port:
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = str(sk.getsockname()[1])
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", port, os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL needs it on this driver
    return subprocess.run(cmd, env=env).returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", type=int, default=2, choices=[2, 5],
                    help="BASELINE.json config: 2 = the headline (pre-train step, default); 5 = the fine-tune step "
                         "(HookNet + Dice + Adam, tools/ssl_finetune.py; defaults resnet18, 64 tile pairs of 256x256)")
    ap.add_argument("--arch", default=None)
    ap.add_argument("--batch", type=int, default=None, help="tile pairs per GPU")
    ap.add_argument("--size", type=int, default=None)
    ap.add_argument("--classes", type=int, default=5, help="config 5: segmentation classes (+1 background channel)")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp16", "fp32"],
                    help="storage / MFMA input type (bf16: BASELINE config 2; fp16: the reference's default --amp dtype)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--stock-batch", type=int, default=0,
                    help="tile pairs of the stock-PyTorch GPU yardstick (0 = skip it, the default: its MIOpen kernel search "
                         "takes minutes; 32 fits eager autograd's memory); it runs with the CPU baseline")
    ap.add_argument("--cpu-budget", type=float, default=20.0)
    ap.add_argument("--no-kernel-timer", action="store_true")
    ap.add_argument("--layer-report", default=None,
                    help="write a per-layer-shape table (time, TFLOP/s, GB/s of every dense and streaming launch of the "
                         "event-timed steps) to this file")
    ap.add_argument("--rendezvous-only", action="store_true",
                    help="start the ranks, form the process group, run the collective probe and exit (launcher test)")
    args = ap.parse_args()
    dflt = {2: ("resnet50", 256, 224), 5: ("resnet18", 64, 256)}[args.config]  # config 5: scripts/bcss.sh:27 (-b 64)
    args.arch = args.arch or dflt[0]
    args.batch = args.batch or dflt[1]
    args.size = args.size or dflt[2]

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: this process has made no GPU call yet; it starts one fresh rank per GPU
        # (torch.distributed.run, the reference's mp.spawn of tools/ssl_train.py:68), relays their output (rank 0
        # prints the JSON line) and exits with their return code
        raise SystemExit(launch_ranks(args.gpus))
    dmabuf_ipc_env()  # the rank path too (a launcher other than launch_ranks may have started this process)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: the launcher and the flag disagree")
    import torch.distributed as dist

    # rehearsal aids for a one-GPU box (RCCL refuses two ranks on one device): MSFWSI_BENCH_DEVICE pins every rank to
    # one card, MSFWSI_BENCH_BACKEND=gloo moves the collectives to the CPU transport.  The driver uses neither.
    backend = os.environ.get("MSFWSI_BENCH_BACKEND", "nccl")
    have_gpu = torch.cuda.is_available()
    if not have_gpu and not (args.rendezvous_only and backend != "nccl"):
        raise SystemExit("bench.py needs an MI355X: the MSF-WSI hot path has no CPU fallback")
    if os.environ.get("MSFWSI_BENCH_DEVICE"):
        local = int(os.environ["MSFWSI_BENCH_DEVICE"])
    if have_gpu:
        torch.cuda.set_device(local)
    if world > 1 or os.environ.get("MSFWSI_FORCE_SYNC", "0") != "0":
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
        from msf_wsi_amd.dist import probe_collectives

        probe_collectives(None, torch.device("cuda", local) if have_gpu else torch.device("cpu"))  # fails loudly
    if args.rendezvous_only:
        if dist.is_initialized():
            dist.barrier()
        if rank == 0:
            print(json.dumps({"rendezvous": "ok", "world": world, "backend": backend if world > 1 else None}),
                  flush=True)
        if dist.is_initialized():
            dist.destroy_process_group()
        return

    from msf_wsi_amd import _lib, kernels as kn
    from msf_wsi_amd.train import PretrainStep, synthetic_batch

    _lib.load()
    dtype = {"bf16": torch.bfloat16, "fp16": torch.float16, "fp32": torch.float32}[args.dtype]
    dev = torch.device("cuda", local)
    if args.config == 5:
        run_finetune(args, world, rank, dev, dtype)
        if dist.is_initialized():
            dist.barrier()
            dist.destroy_process_group()
        return
    model = build(args.arch, dev, rank)
    ts = PretrainStep(model, lr=1e-3, global_batch=args.batch * world, dtype=dtype, arch=args.arch)
    ts.engine.recompute = os.environ.get("MSFWSI_RECOMPUTE", "auto")
    batch = synthetic_batch(args.batch, args.size, 16, seed=rank, device=dev)

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        ts.step(batch)
    timer = None if args.no_kernel_timer else kn.KernelTimer(streams=True)
    sync()
    # One rank runs the step on three HIP streams (target views on two, context passes on a third: engine.Engine).  An
    # event pair around a launch on a stream that SHARES the chip measures that launch plus its queueing behind the other
    # streams' kernels (measured: the summed "durations" exceed the step time several times over) -- useless for a
    # roofline.  In that case the timed region carries NO events, and the roofline comes from ONE EXTRA step right after
    # it, on one stream, whose launches are bracketed as before: the same kernels on the same tensors, each alone on the
    # chip.  `value` / `ms_per_step` are of the timed region only.
    multi = ",dual-stream" in ts.engine.last_plan  # decided by the engine during the warm-up steps
    # otherwise (one stream; more than one rank): the per-launch HIP events (two per launch, ~7 500 per step) cost ~3 % of
    # a step they bracket and are recorded on the LAST timed step only
    timed_from = args.steps if multi else max(0, args.steps - 1)
    t0 = time.perf_counter()
    for i in range(args.steps):
        kn.TIMER = timer if i >= timed_from else None
        loss = ts.step(batch)
    sync()
    dt = time.perf_counter() - t0
    kn.TIMER = None
    run_plan = ts.engine.last_plan
    if timer is not None and ",dual-stream" in run_plan and not multi:
        # fewer than two warm-up steps: the engine switched to several streams INSIDE the timed region, so the last step's
        # event pairs are of shared streams -- discard them and measure the extra step instead
        timer, multi = kn.KernelTimer(streams=True), True
    run_collectives = ts.engine.collectives_last_step + ts.reducer.launches_last_step
    peak_mem = torch.cuda.max_memory_allocated() / 2 ** 30
    peak_reserved = torch.cuda.max_memory_reserved() / 2 ** 30
    ev_nst, ev_seconds = None, None
    if timer is not None and multi:
        torch.cuda.empty_cache()  # the side streams' cached blocks go back before one stream needs room for every pass
        keep = ts.engine.dual_stream
        ts.engine.dual_stream = False
        ts.step(batch)  # un-bracketed: the main stream's pool grows back to hold every pass (seconds of hipMalloc)
        kn.TIMER = timer
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        ts.step(batch)
        torch.cuda.synchronize()
        ev_nst, ev_seconds = 1, time.perf_counter() - t1
        kn.TIMER = None
        ts.engine.dual_stream = keep
    # A number measured on a recompute plan (a pass run features-only and re-run before its backward: +16 % / +32 %
    # time) is not the configuration the metric names unless the caller asked for it: fail loudly instead of printing it
    # (several ranks: the plan is COLLECTIVE -- the most conservative rank decides -- and a first contact with real peers must
    #  produce a line, not die on policy: the line is printed with the plan labelled)
    if "recompute:" in run_plan and "MSFWSI_RECOMPUTE" not in os.environ and world == 1:
        raise SystemExit(f"bench.py: the memory plan of this run fell back to '{run_plan}' (not enough HBM to "
                         f"keep every activation at {args.batch} tile pairs per GPU); set MSFWSI_RECOMPUTE=auto to accept "
                         f"a number measured with recomputation, or lower --batch")
    plan8 = None
    if world == 1 and ts.engine.last_shape is not None:
        torch.cuda.empty_cache()
        Bs, Ks, sk = ts.engine.last_shape
        plan8 = ts.engine.plan_preview(Bs, Ks, sk, dev, 8)  # with the 4 GiB RCCL reserve of a multi-rank run
    tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
    replicas_equal = None
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        # the replicas started from DIFFERENT seeds (build) and must have ended the run with equal weights: the checksum of the
        # weights every kernel reads (the 16-bit copies where they exist, else the fp32 masters), MIN and MAX over the ranks
        chk = torch.stack([(w16 if w16 is not None else w).double().abs().sum()
                           for w, w16 in zip(ts.flats.w, ts.flats.w16)])
        lo, hi = chk.clone(), chk.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        replicas_equal = bool(torch.equal(lo, hi))
    dt = float(tmax.item())

    if rank == 0:
        pairs = args.batch * world * args.steps
        out = {
            "metric": "tile-pairs/sec per pretrain step", "value": round(pairs / dt, 3), "unit": "tile-pairs/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * dt / args.steps, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": f"{args.arch} dual-stream MSF-WSI pre-train step, {args.batch} tile pairs/GPU "
                                   f"(34 image passes of {args.size}x{args.size}x3 each), {args.dtype}, "
                                   f"Adam + GradScaler, SyncBN+DP over {world} GPU(s)",
                       "arch": args.arch, "batch_per_gpu": args.batch, "global_batch": args.batch * world,
                       "image_size": args.size, "parallelism": f"dp{world}",
                       "images_per_s": round(34 * pairs / dt, 1), "loss": float(loss),
                       "peak_mem_GiB": round(peak_mem, 1), "peak_reserved_GiB": round(peak_reserved, 1),
                       "step_TFLOP_per_s_algorithmic": round(FLOP_PER_PAIR.get(args.arch, 0) * pairs / dt / 1e12, 1),
                       "recompute_plan": run_plan, "plan_at_8_ranks": plan8,
                       # a multi-rank line measured on a recompute plan says so here (+16 % / +32 % time per re-run pass)
                       "recompute_fallback": "recompute:" in run_plan and "MSFWSI_RECOMPUTE" not in os.environ,
                       # device memory the communicators took outside torch's pool when they were created and probed
                       # (Engine.prepare_multirank: mem_get_info before / after)
                       "comm_GiB": round(ts.engine.comm_bytes / 2 ** 30, 2),
                       # several ranks: the weights' checksum is the same on every rank after the run (the ranks were seeded
                       # differently; PretrainStep's rank-0 broadcast and the gradient exchange keep them equal)
                       "replicas_equal": replicas_equal,
                       # SyncBatchNorm exchanges (+ the plan) of the last step and the gradient exchange's messages
                       "collectives_per_step": run_collectives,
                       "gradient_messages_per_step": ts.reducer.launches_last_step,
                       "gradient_bytes_per_step": ts.reducer.bytes_last_step,
                       # sharded: every bucket reduce-scattered, Adam on 1/world of it, updated weights all-gathered (fp32
                       # masters of the encoders, the 16-bit copy alone for the fuser heads)
                       "optimizer": ("sharded (reduce-scatter, Adam on 1/%d, all-gather)" % max(1, world)
                                     if ts.reducer.sharding else "replicated (all-reduce)" if ts.reducer.active else "single rank")},
        }
        if timer is not None and args.layer_report:
            rows = sorted(timer.by_shape().items(), key=lambda kv: -kv[1][1])
            nst = ev_nst if ev_nst is not None else args.steps - timed_from
            with open(args.layer_report, "w") as f:
                f.write(f"# per event-timed step ({nst} steps averaged"
                        + (", the extra ONE-STREAM step after the timed region" if ev_nst is not None else "")
                        + "); ms, TFLOP/s, GB/s are algorithmic\n")
                f.write("kind\tms_per_step\tlaunches_per_step\tTFLOP/s\tGB/s\tshape\tsymbol\n")
                for (kind, shape, sym), (n, sec, fl, by) in rows:
                    f.write(f"{kind}\t{1e3 * sec / nst:.3f}\t{n / nst:.1f}\t{fl / sec / 1e12:.1f}\t{by / sec / 1e9:.0f}\t"
                            f"{shape}\t{sym if kind != 'stream' else ''}\n")
        if timer is not None:
            area = (args.size / 224.0) ** 2  # SURVEY 8(d)'s per-pair figures are for 224 x 224 tiles
            step_bytes = (ACT_BYTES_PER_PAIR_16BIT.get(args.arch, 0) * (2 if args.dtype == "fp32" else 1) * args.batch * area
                          + ADAM_BYTES_PER_STEP.get(args.arch, 0))
            # the committed counter summary is of ONE workload (config 2 as the default command runs it): replayed
            # only beside that workload, never beside another architecture / batch / tile size / dtype
            profiled = (args.arch, args.batch, args.size, args.dtype) == ("resnet50", 256, 224, "bf16")
            out["roofline"] = build_roofline(timer, args, dt, timed_from, step_bytes,
                                             FLOP_PER_PAIR.get(args.arch, 0) * args.batch * area, use_pmc=profiled,
                                             nst=ev_nst, ev_seconds=ev_seconds)
            out["roofline"]["concurrent_streams_in_timed_region"] = 3 if "+context-stream" in run_plan else (2 if multi else 1)
            out["roofline"]["measured_on"] = (
                "one extra step right after the timed region (and one un-bracketed step that regrows the allocator pool), on ONE "
                "stream (the timed region runs three streams, where "
                "an event pair brackets queueing, not a kernel); ms of that step: %.1f" % (1e3 * ev_seconds)
                if multi else "the last step of the timed region")
        if world == 1 and not args.no_cpu_baseline:
            del ts, model, batch
            torch.cuda.empty_cache()
            # the CPUs this process may really use (cgroup quota: 16 on a GPU box that shows 256, msf_wsi_amd/hostcpu.py)
            from msf_wsi_amd.hostcpu import usable_cpus

            threads = usable_cpus()
            out["cpu_baseline"] = cpu_baseline(args.arch, args.size, args.cpu_budget, threads)
            if args.stock_batch > 0:
                try:
                    out["stock_gpu_baseline"] = stock_gpu_baseline(args.arch, args.size, args.stock_batch)
                except Exception as e:  # noqa: BLE001 -- a yardstick must never take the bench line down
                    out["stock_gpu_baseline"] = {"value": None, "error": f"{type(e).__name__}: {e}"[:300]}
        print(json.dumps(out), flush=True)
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
