"""micro-benchmark of the dense kernels on ResNet-50 layer shapes (diagnostic, not part of the product)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from msf_wsi_amd import kernels as kn  # noqa: E402

from msf_wsi_amd import _lib as _L  # noqa: E402

_L.load().msfwsi_set_tuning(1, int(os.environ.get("TUNE_FAST", "1")))
_L.load().msfwsi_set_tuning(0, int(os.environ.get("TUNE_BIG", "1024")))
N = int(os.environ.get("NIMG", "1024"))
REP = int(os.environ.get("REP", "5"))
ONLY = os.environ.get("ONLY", "")
HALO = os.environ.get("HALO", "1") != "0"
dt = torch.bfloat16
SHAPES = [  # name, H, C, K, R, stride, pro
    ("l1.conv1 1x1 256->64", 56, 256, 64, 1, 1, False),
    ("l1.conv2 3x3 64->64 pro", 56, 64, 64, 3, 1, True),
    ("l1.conv2 3x3 64->64 nopro", 56, 64, 64, 3, 1, False),
    ("l1.conv3 1x1 64->256 pro", 56, 64, 256, 1, 1, True),
    ("l2.conv2 3x3 128->128 pro", 28, 128, 128, 3, 1, True),
    ("l2.conv3 1x1 128->512 pro", 28, 128, 512, 1, 1, True),
    ("l3.conv2 3x3 256->256 pro", 14, 256, 256, 3, 1, True),
    ("l3.conv2 3x3 256->256 nopro", 14, 256, 256, 3, 1, False),
    ("l3.conv1 1x1 1024->256", 14, 1024, 256, 1, 1, False),
    ("l4.conv2 3x3 512->512 pro", 7, 512, 512, 3, 1, True),
    ("l2.conv2 3x3 128->128 nopro", 28, 128, 128, 3, 1, False),
    ("l4.conv2 3x3 512->512 nopro", 7, 512, 512, 3, 1, False),
    ("l1.conv3 1x1 64->256 nopro", 56, 64, 256, 1, 1, False),
    ("l2.conv1 1x1 512->128", 28, 512, 128, 1, 1, False),
    ("l2.conv3 1x1 128->512 nopro", 28, 128, 512, 1, 1, False),
    ("l3.conv3 1x1 256->1024 nopro", 14, 256, 1024, 1, 1, False),
    ("l4.conv1 1x1 2048->512", 7, 2048, 512, 1, 1, False),
    ("l4.conv3 1x1 512->2048 nopro", 7, 512, 2048, 1, 1, False),
]


def timeit(fn):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(REP):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / REP


for name, H, C, K, R, st, pro in SHAPES:
    if ONLY and ONLY not in name:
        continue
    d = kn.conv_desc(dt, N, H, H, C, K, R, R, st, R // 2)
    x = torch.randn(N, H, H, C, device="cuda").to(dt)
    w = (torch.randn(K, R, R, C, device="cuda") * 0.05).to(dt)
    y = torch.empty(N, d.P, d.Q, K, dtype=dt, device="cuda")
    dy = torch.randn(N, d.P, d.Q, K, device="cuda").to(dt)
    dx = torch.empty_like(x)
    dw = torch.zeros(K, R, R, C, device="cuda")
    sc, sh = torch.rand(C, device="cuda") + 0.5, torch.randn(C, device="cuda") * 0.1
    stats = kn.new_stats(K)
    M = N * d.P * d.Q
    fl = 2.0 * M * K * R * R * C
    by = 2.0 * (x.numel() + y.numel() + w.numel())
    p = (sc, sh) if pro else None
    if HALO and R == 3 and not pro and kn.conv3x3_supported(d):
        t_f = timeit(lambda: kn.conv3x3_fwd(d, x, w, y, stats=stats))
        t_d = timeit(lambda: kn.conv3x3_dgrad(d, dy, w, dx))
    else:
        t_f = timeit(lambda: kn.conv_fwd(d, x, w, y, pro=p, stats=stats))
        t_d = timeit(lambda: kn.conv_dgrad(d, dy, w, dx))
    t_w = timeit(lambda: kn.conv_wgrad(d, x, dy, dw, pro=p, target_blocks=int(os.environ.get("WG_BLOCKS", "0"))))
    if os.environ.get("POST", "0") != "0" and R == 1:
        ident = torch.randn_like(y)
        ps, pb = torch.rand(K, device="cuda") + 0.5, torch.randn(K, device="cuda")
        bits = kn.gate_bytes(M, K, dt)
        t_p = timeit(lambda: kn.conv_fwd_post(d, x, w, y, ps, pb, ident=ident, relu=True, gate_out=bits))
        t_p0 = timeit(lambda: kn.conv_fwd_post(d, x, w, y, ps, pb, ident=None, relu=True))
        byp = 2.0 * (x.numel() + 2 * y.numel())
        print(f"{name:30s} fwd_post+ident+bits {t_p:7.3f} ms {byp / t_p / 1e6:7.1f} GB/s | fwd_post plain {t_p0:7.3f} ms {by / t_p0 / 1e6:7.1f} GB/s | fwd+stats {t_f:7.3f} ms {by / t_f / 1e6:7.1f} GB/s", flush=True)
    if os.environ.get("FUSED", "0") != "0":
        # the epilogue the engine really uses: residual add + ReLU gate of the producer + BatchNorm-backward sums
        resid = torch.randn_like(dx)
        mc = torch.randn_like(dx)
        one, zero = torch.ones(C, device="cuda"), torch.zeros(C, device="cuda")
        sums = kn.new_stats(C, 2, "cuda")
        t_df = timeit(lambda: kn.conv_dgrad(d, dy, w, dx, resid=resid, mask=(mc, one, zero), sums=sums))
        byf = 2.0 * (dy.numel() + 3 * dx.numel())
        print(f"{name:30s} fused dgrad (resid+gate+sums) {t_df:7.3f} ms {byf / t_df / 1e6:7.1f} GB/s | plain dgrad {2.0 * (dy.numel() + dx.numel()) / t_d / 1e6:7.1f} GB/s | "
              f"fwd+stats {by / t_f / 1e6:7.1f} GB/s | wgrad {2.0 * (x.numel() + dy.numel()) / t_w / 1e6:7.1f} GB/s", flush=True)
    print(f"{name:30s} M={M:9d} algMB x {x.numel() * 2 / 1e6:.0f} y {y.numel() * 2 / 1e6:.0f} | fwd {t_f:7.3f} ms {fl / t_f / 1e9:7.1f} TF {by / t_f / 1e6:7.1f} GB/s | "
          f"dgrad {t_d:7.3f} ms {fl / t_d / 1e9:7.1f} TF | wgrad {t_w:7.3f} ms {fl / t_w / 1e9:7.1f} TF", flush=True)
