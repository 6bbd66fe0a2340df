"""GPU box: the 64 -> 64 3x3 layers of the bench workload (N = 4096 target tiles of 56 x 56): weights-stationary
persistent kernel against the gather kernels.   python tools/c3_bench.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from msf_wsi_amd import _lib, kernels as kn  # noqa: E402


def timed(fn):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 5


def main():
    lib = _lib.load()
    dt = torch.bfloat16
    for N in (4096, 256):
        H = W = 56
        Cc = 64
        d = kn.conv_desc(dt, N, H, W, Cc, Cc, 3, 3, 1, 1)
        x = torch.randn(N * H * W * Cc, device="cuda").to(dt)
        w = (torch.randn(Cc * 9 * Cc, device="cuda") * 0.05).to(dt)
        y = torch.empty_like(x)
        c = torch.randn(N * H * W * Cc, device="cuda").to(dt)
        sc, sh = torch.rand(Cc, device="cuda"), torch.zeros(Cc, device="cuda")
        stats, sums = kn.new_stats(Cc), kn.new_stats(Cc)
        flop = 2.0 * N * H * W * Cc * 9 * Cc
        line = f"N{N} 56x56 C64->K64 3x3: "
        for on in (0, 1):
            lib.msfwsi_set_tuning(9, on)
            f = timed(lambda: kn.conv3x3_fwd(d, x, w, y, stats=stats))
            g = timed(lambda: kn.conv3x3_dgrad(d, x, w, y, mask=(c, sc, sh), sums=sums))
            line += f"  stationary={on}: fwd {f:.3f} ms {flop / f / 1e9:5.0f} TF, dgrad(gated) {g:.3f} ms {flop / g / 1e9:5.0f} TF"
        f = timed(lambda: kn.conv_fwd(d, x, w, y, stats=stats))
        line += f"  gather fwd {f:.3f} ms {flop / f / 1e9:5.0f} TF"
        dw = torch.zeros(Cc * 9 * Cc, device="cuda")
        for on in (0, 1):
            lib.msfwsi_set_tuning(10, on)
            g = timed(lambda: kn.conv_wgrad(d, x, c, dw))
            gp = timed(lambda: kn.conv_wgrad(d, x, c, dw, pro=(sc, sh))) if on else float("nan")
            line += f"  wgrad os={on}: {g:.3f} ms {flop / g / 1e9:5.0f} TF (fused prologue {gp:.3f} ms)"
        lib.msfwsi_set_tuning(10, 1)
        print(line, flush=True)
    lib.msfwsi_set_tuning(9, 1)
    # the stem on its space-to-depth input: [N][112][112][16] -> 64 channels, 4x4 / stride 1 / pad 2
    N, H2 = 4096, 112
    xs = torch.randn(N, H2, H2, 16, device="cuda").to(dt)
    w2 = (torch.randn(64, 4, 4, 16, device="cuda") * 0.05).to(dt)
    c = torch.empty(N, H2, H2, 64, device="cuda", dtype=dt)
    stats = kn.new_stats(64)
    flop = 2.0 * N * H2 * H2 * 64 * 256
    line = "stem N4096 112x112 C16->K64 4x4: "
    for on in (0, 1):
        lib.msfwsi_set_tuning(12, on)
        f = timed(lambda: kn.stem_conv_fwd(xs, w2, c, stats, 4, 4, 1, 2, P=H2, Q=H2))
        gb = (xs.numel() + c.numel()) * 2 / f / 1e6
        line += f"  stationary={on}: {f:.3f} ms {flop / f / 1e9:5.0f} TF {gb:5.0f} GB/s"
    print(line, flush=True)
    lib.msfwsi_set_tuning(12, 1)
    d = kn.conv_desc(dt, N, H2, H2, 16, 64, 4, 4, 1, 2)
    d = type(d)(d.dtype, N, H2, H2, 16, H2, H2, 64, 4, 4, 1, 2)
    dw = torch.zeros(64 * 256, device="cuda")
    line = "stem wgrad: "
    for on in (0, 1):
        lib.msfwsi_set_tuning(12, on)
        f = timed(lambda: kn.conv_wgrad(d, xs, c, dw))
        line += f"  stationary={on}: {f:.3f} ms {flop / f / 1e9:5.0f} TF"
    print(line, flush=True)
    lib.msfwsi_set_tuning(12, 1)


if __name__ == "__main__":
    main()
