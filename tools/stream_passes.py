"""GPU box: which stand-alone streaming passes does one full-size step still run?  Wraps the kernels.py entry points of the
elementwise family, runs two steps of BASELINE config 2 and prints, per (kernel, tensor shape), the launches per step and the
bytes they move (tensors read + written once).   python tools/stream_passes.py"""
import collections
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from msf_wsi_amd import kernels as kn  # noqa: E402
from msf_wsi_amd.train import PretrainStep, synthetic_batch  # noqa: E402

LOG = collections.defaultdict(lambda: [0, 0.0])
ON = [False]


def wrap(name, nbytes):
    real = getattr(kn, name)

    def f(*a, **k):
        if ON[0]:
            first = next(t for t in a if isinstance(t, torch.Tensor))
            key = (name, tuple(first.shape))
            LOG[key][0] += 1
            LOG[key][1] += nbytes(a, k)
        return real(*a, **k)

    setattr(kn, name, f)


def sz(t):
    return 0 if t is None else t.numel() * t.element_size()


wrap("bn_act", lambda a, k: sz(a[0]) + sz(a[3]) + sz(k.get("ident")))
wrap("bn_act_sum", lambda a, k: sz(a[0]) + sz(a[3]))
wrap("bn_bwd_apply", lambda a, k: sz(a[0]) + sz(a[1]) + sz(a[5]))
wrap("pixel_stride", lambda a, k: sz(a[0]) + sz(a[1]))
wrap("gap_fwd", lambda a, k: sz(a[0]))
wrap("gap_fwd_stride2", lambda a, k: sz(a[0]) + sz(a[2]))
wrap("block_end_bwd", lambda a, k: sz(a[0]) + 2 * sz(a[1]))
wrap("act_bwd_reduce", lambda a, k: 2 * sz(a[0]) + sz(a[1]))
wrap("stem_pool_fwd", lambda a, k: sz(a[0]) + sz(a[3]) + sz(a[4]))
wrap("stem_pool_bwd", lambda a, k: sz(a[0]) + sz(a[1]) + sz(a[2]) + sz(a[5]))

model = bench.build("resnet50", torch.device("cuda", 0))
ts = PretrainStep(model, lr=1e-3, global_batch=256, dtype=torch.bfloat16, arch="resnet50")
batch = synthetic_batch(256, 224, 16, seed=0, device="cuda")
ts.step(batch)
ts.step(batch)
ON[0] = True
ts.step(batch)
torch.cuda.synchronize()
tot = 0.0
for (name, shape), (n, b) in sorted(LOG.items(), key=lambda kv: -kv[1][1]):
    tot += b
    if b > 2e8:
        print(f"{name:18s} {str(shape):28s} {n:4d} launches  {b / 1e9:8.2f} GB/step")
print(f"total {tot / 1e9:.1f} GB/step")
