"""GPU box: weight-gradient kernel A/B on the deep-layer shapes of the bench workload (N = 4096 target tiles).
    python tools/wgrad_bench.py      -> per shape: ms and TFLOP/s with the 256x256 sixteen-wave tile and without"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from msf_wsi_amd import _lib, kernels as kn  # noqa: E402

SHAPES = [  # N, H, W, C, K, R, stride   (N = int(os.environ.get("WGRAD_BENCH_N", 4096)) replaces the first field)
    (4096, 14, 14, 256, 256, 3, 1), (4096, 7, 7, 512, 512, 3, 1), (4096, 14, 14, 1024, 256, 1, 1),
    (4096, 14, 14, 256, 1024, 1, 1), (4096, 7, 7, 512, 2048, 1, 1), (4096, 7, 7, 2048, 512, 1, 1),
    (4096, 14, 14, 512, 1024, 1, 1), (4096, 28, 28, 256, 256, 3, 2), (4096, 28, 28, 256, 512, 1, 1),
    (4096, 28, 28, 512, 256, 1, 1),
]


def main():
    lib = _lib.load()
    dt = torch.bfloat16
    for N, H, W, Cc, K, R, st in SHAPES:
        N = int(os.environ.get("WGRAD_BENCH_N", N))
        pad = R // 2
        d = kn.conv_desc(dt, N, H, W, Cc, K, R, R, st, pad)
        x = torch.randn(N * H * W * Cc, device="cuda").to(dt)
        dy = (torch.randn(N * d.P * d.Q * K, device="cuda") * 0.05).to(dt)
        dw = torch.zeros(K * R * R * Cc, device="cuda")
        flop = 2.0 * N * d.P * d.Q * K * R * R * Cc
        line = f"N{N} {H}x{W} C{Cc}->K{K} {R}x{R}/s{st}: "
        for big in (0, 1):
            lib.msfwsi_set_tuning(6, big)
            for _ in range(2):
                kn.conv_wgrad(d, x, dy, dw)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                kn.conv_wgrad(d, x, dy, dw)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 5
            line += f"  big={big}: {ms:.3f} ms {flop / ms / 1e9:6.0f} TF"
        print(line, flush=True)
        del x, dy, dw
    lib.msfwsi_set_tuning(6, 1)


if __name__ == "__main__":
    main()
