"""GPU box: is the register-staged BatchNorm-prologue weight gradient (XPRO) competitive with
bn_act + the linear-DMA weight gradient on the HBM-bound 1x1 shapes (Gram matrices and M = g^T a2)?
    python tools/xpro_bench.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from msf_wsi_amd import _lib, kernels as kn  # noqa: E402

SHAPES = [  # N, HW, C (operand), K (dy)   -- Gram: K == C with dy = x
    (4096, 56, 64, 64), (4096, 28, 128, 128), (4096, 14, 256, 256),
    (4096, 56, 64, 256), (4096, 28, 128, 512), (4096, 14, 256, 1024),
]


def timed(fn, n=5):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def main():
    _lib.load()
    dt = torch.bfloat16
    for N, HW, Cc, K in SHAPES:
        d = kn.conv_desc(dt, N, HW, HW, Cc, K, 1, 1, 1, 0)
        M = N * HW * HW
        c = torch.randn(M, Cc, device="cuda").to(dt)
        a = torch.empty_like(c)
        dy = a if K == Cc else (torch.randn(M, K, device="cuda") * 0.05).to(dt)
        dw = torch.zeros(K * Cc, device="cuda")
        sc, sh = torch.rand(Cc, device="cuda") + 0.5, torch.randn(Cc, device="cuda") * 0.1
        t_act = timed(lambda: kn.bn_act(c, sc, sh, a, relu=True))
        t_lin = timed(lambda: kn.conv_wgrad(d, a, dy, dw))
        t_pro = timed(lambda: kn.conv_wgrad(d, c, dy, dw, pro=(sc, sh)))
        gb = (M * Cc * 2 + (0 if K == Cc else M * K * 2)) / 1e9
        print(f"N{N} {HW}x{HW} C{Cc} K{K}: bn_act {t_act:.3f} ms, wgrad(DMA) {t_lin:.3f} ms ({gb / t_lin * 1e3:.0f} GB/s), "
              f"wgrad(prologue) {t_pro:.3f} ms; pair {t_act + t_lin:.3f} vs fused(+write-back est.) "
              f"{t_pro + M * Cc * 2 / 5e9 * 1e-3 * 0:.3f}", flush=True)


if __name__ == "__main__":
    main()
