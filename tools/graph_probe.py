"""GPU box experiment: the whole fused pre-train step captured ONCE into a HIP graph (torch.cuda.CUDAGraph) and replayed --
does removing ~3 200 host-side launches per step (Python + ctypes, three streams) move the step time?
    python tools/graph_probe.py [steps]        (BASELINE config 2: ResNet-50, 256 tile pairs, bf16)"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import build  # noqa: E402
from msf_wsi_amd.train import PretrainStep, synthetic_batch  # noqa: E402


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    arch, B = os.environ.get("PROBE_ARCH", "resnet50"), int(os.environ.get("PROBE_BATCH", "256"))
    dev = torch.device("cuda", 0)
    model = build(arch, dev)
    ts = PretrainStep(model, lr=1e-3, global_batch=B, dtype=torch.bfloat16, arch=arch)
    (c1, c2), (t1, t2), idx = synthetic_batch(B, 224, 16, seed=0, device=dev)
    batch = ((c1, c2), (t1, t2), [i.to(dev) for i in idx])  # device-resident indices: no host copy inside the capture
    for _ in range(3):
        ts.step(batch)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        ts.step(batch)
    torch.cuda.synchronize()
    eager = (time.perf_counter() - t0) / steps
    print(f"eager: {1e3 * eager:.1f} ms/step, plan {ts.engine.last_plan}", flush=True)
    ts.engine.recompute = "off"  # the plan is known: no memory query inside the capture
    torch.cuda.empty_cache()
    g = torch.cuda.CUDAGraph()
    t0 = time.perf_counter()
    with torch.cuda.graph(g, capture_error_mode="relaxed"):
        loss = ts.step(batch)
    torch.cuda.synchronize()
    print(f"captured in {time.perf_counter() - t0:.1f} s; reserved {torch.cuda.memory_reserved() / 2**30:.1f} GiB", flush=True)
    for _ in range(2):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        g.replay()
    torch.cuda.synchronize()
    graph = (time.perf_counter() - t0) / steps
    print(f"graph replay: {1e3 * graph:.1f} ms/step (eager {1e3 * eager:.1f}); loss {float(loss):.5f}, Adam steps {ts.t}", flush=True)


if __name__ == "__main__":
    main()
