"""GPU box: micro-benchmarks of single launches at the shapes of BASELINE config 2 (ResNet-50, 256 tile pairs: N = 4096
target tiles), through the same C-ABI entry points the engine uses.  One line per case: ms per launch, algorithmic GB/s
and TFLOP/s.     python tools/kbench.py [case ...]        (MSFWSI_LIB=other.so for an A/B of two builds)

cases: epi3 (strided-residual 1x1 input gradients), s2 (stride-2 3x3 input gradients), wide (short-k 1x1 launches with
wide outputs), pool (stem max-pool forward / backward), fuser (18432-wide Linear layers at 512 rows), dma (plain 1x1)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from msf_wsi_amd import _lib, kernels as kn  # noqa: E402

DT = torch.bfloat16
NIMG = int(os.environ.get("KBENCH_N", 4096))
ITERS = int(os.environ.get("KBENCH_ITERS", 6))


def rnd(*shape, scale=1.0, dtype=DT):
    return (torch.randn(*shape, device="cuda") * scale).to(dtype)


def timeit(fn, iters=ITERS, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def report(name, ms, nbytes, flop=0.0):
    print(f"{name:78s} {ms:8.3f} ms  {nbytes / ms / 1e6:7.0f} GB/s  {flop / ms / 1e9:7.0f} TF", flush=True)


def case_epi3():
    """conv1 (1x1) input gradient of the three strided Bottlenecks: gate bits + sums + the low-resolution residual"""
    for H, C, K in ((56, 256, 128), (28, 512, 256), (14, 1024, 512)):
        N = NIMG
        d = kn.conv_desc(DT, N, H, H, C, K, 1, 1, 1, 0)
        M = N * H * H
        dy = rnd(M, K, scale=0.05)
        w = rnd(K, C, scale=0.05)
        dx = torch.empty(M, C, dtype=DT, device="cuda")
        lo = rnd(N * (H // 2) * (H // 2), C, scale=0.05)
        full = rnd(M, C, scale=0.05)
        bits = torch.randint(0, 256, (kn.gate_numel(M, C, DT),), dtype=torch.uint8, device="cuda")
        gapg = rnd(N, C, scale=0.05)
        base = (M * K + M * C) * 2 + M * C // 8
        flop = 2.0 * M * K * C
        for label, kw, extra in (
            ("bits+sums, no residual", dict(), 0),
            ("bits+sums+lowres residual (engine)", dict(resid=lo, resid_stride=2), lo.numel() * 2),
            ("bits+sums+full residual", dict(resid=full), full.numel() * 2),
            ("bits+sums+gap+lowres residual", dict(resid=lo, resid_stride=2, gapg=gapg, gap_scale=1.0 / (H * H)), lo.numel() * 2),
        ):
            def run():
                sums = kn.new_stats(C, 2, "cuda")
                kn.conv_dgrad(d, dy, w, dx, mask_bits=bits, sums=sums, **kw)
            report(f"epi3 {H}x{H} dY{K}->dX{C}: {label}", timeit(run), base + extra, flop)
        del dy, dx, lo, full, bits


def case_s2():
    """conv2 (3x3 / stride 2) input gradients of the strided Bottlenecks, gated by the producer's BatchNorm+ReLU"""
    for H, C in ((56, 128), (28, 256), (14, 512)):
        N = NIMG
        d = kn.conv_desc(DT, N, H, H, C, C, 3, 3, 2, 1)
        M, Mo = N * H * H, N * d.P * d.Q
        dy = rnd(Mo, C, scale=0.05)
        w = rnd(C * 9, C, scale=0.05)
        dx = torch.empty(M, C, dtype=DT, device="cuda")
        c = rnd(M, C)
        sc, sh = torch.ones(C, device="cuda"), torch.zeros(C, device="cuda")
        nbytes = (Mo * C + 2 * M * C) * 2
        flop = 2.0 * Mo * 9 * C * C

        def run():
            sums = kn.new_stats(C, 2, "cuda")
            kn.conv_dgrad(d, dy, w, dx, mask=(c, sc, sh), sums=sums)
        report(f"s2   {H}x{H} C{C} 3x3/s2 dgrad (+gate, sums)", timeit(run), nbytes, flop)
        del dy, dx, c


def case_wide():
    """short-k 1x1 launches with a wide output: conv3 forward with its fused tail, conv1 input gradient"""
    for H, Cn, Kw in ((14, 256, 1024), (28, 128, 512), (7, 512, 2048), (56, 64, 256)):
        N = NIMG
        M = N * H * H
        # forward conv3: a2 [M][Cn] -> y [M][Kw], BatchNorm apply + identity + ReLU + gate bits
        d = kn.conv_desc(DT, N, H, H, Cn, Kw, 1, 1, 1, 0)
        a = rnd(M, Cn)
        w = rnd(Kw, Cn, scale=0.05)
        y = torch.empty(M, Kw, dtype=DT, device="cuda")
        ident = rnd(M, Kw)
        ps, pb = torch.ones(Kw, device="cuda"), torch.zeros(Kw, device="cuda")
        bits = kn.gate_bytes(M, Kw, DT, "cuda")
        nb = (M * Cn + 2 * M * Kw) * 2 + M * Kw // 8
        report(f"wide {H}x{H} fwd C{Cn}->K{Kw} post (bn+ident+relu+bits)",
               timeit(lambda: kn.conv_fwd_post(d, a, w, y, ps, pb, ident=ident, relu=True, gate_out=bits)), nb,
               2.0 * M * Cn * Kw)
        # input gradient of conv1 (Kw -> Cn): dY [M][Cn] -> dX [M][Kw], residual + gate bits + sums
        d1 = kn.conv_desc(DT, N, H, H, Kw, Cn, 1, 1, 1, 0)
        w1 = rnd(Cn, Kw, scale=0.05)

        def run():
            sums = kn.new_stats(Kw, 2, "cuda")
            kn.conv_dgrad(d1, a, w1, y, resid=ident, mask_bits=bits, sums=sums)
        report(f"wide {H}x{H} dgrad dY{Cn}->dX{Kw} (+resid, bits, sums)", timeit(run), nb, 2.0 * M * Cn * Kw)
        del a, y, ident, bits


def case_panel():
    """the activation-stationary kernels (csrc/panel.hip) at the shapes of case `wide` / `epi3`: forward tail with and
    without the fused BatchNorm+ReLU prologue, conv1 input gradient with and without the fused BatchNorm backward"""
    only = [int(v) for v in os.environ.get("KBENCH_PANEL_H", "").split(",") if v]  # e.g. KBENCH_PANEL_H=14: that shape alone
    for H, Cn, Kw in ((14, 256, 1024), (28, 128, 512), (7, 512, 2048), (56, 64, 256)):
        if only and H not in only:
            continue
        N = NIMG
        M = N * H * H
        d = kn.conv_desc(DT, N, H, H, Cn, Kw, 1, 1, 1, 0)
        a = rnd(M, Cn)
        w = rnd(Kw, Cn, scale=0.05)
        wpk = kn.panel_pack_weights(w, torch.empty_like(w), Kw, Cn, Cn, 1)
        y = torch.empty(M, Kw, dtype=DT, device="cuda")
        ident = rnd(M, Kw)
        ps, pb = torch.ones(Kw, device="cuda"), torch.zeros(Kw, device="cuda")
        sc, sh = torch.ones(Cn, device="cuda"), torch.zeros(Cn, device="cuda")
        bits = kn.gate_bytes(M, Kw, DT, "cuda")
        nb = (M * Cn + 2 * M * Kw) * 2 + M * Kw // 8
        fl = 2.0 * M * Cn * Kw
        report(f"panel {H}x{H} fwd C{Cn}->K{Kw} post (bn+ident+relu+bits)",
               timeit(lambda: kn.panel_fwd_post(d, a, wpk, y, ps, pb, ident=ident, relu=True, gate_out=bits)), nb, fl)
        report(f"panel {H}x{H} fwd C{Cn}->K{Kw} post + bn2/relu prologue",
               timeit(lambda: kn.panel_fwd_post(d, a, wpk, y, ps, pb, pro=(sc, sh), ident=ident, relu=True, gate_out=bits)), nb, fl)
        # conv1 input gradient: dY [M][Cn] -> dX [M][Kw]
        d1 = kn.conv_desc(DT, N, H, H, Kw, Cn, 1, 1, 1, 0)
        w1 = rnd(Cn, Kw, scale=0.05)
        wpk1 = kn.panel_pack_weights(w1, torch.empty_like(w1), Kw, Cn, 1, Kw)
        c1 = rnd(M, Cn)
        dc = torch.empty(M, Cn, dtype=DT, device="cuda")

        def run(bn):
            sums = kn.new_stats(Kw, 2, "cuda")
            kn.panel_dgrad(d1, a, wpk1, y, bnbwd=(c1, sc, sh, sh) if bn else None, dc_out=dc if bn else None, resid=ident,
                           mask_bits=bits, sums=sums)
        report(f"panel {H}x{H} dgrad dY{Cn}->dX{Kw} (+resid, bits, sums)", timeit(lambda: run(False)), nb, fl)
        report(f"panel {H}x{H} dgrad dY{Cn}->dX{Kw} (+resid, bits, sums) + bn1 backward, dc written",
               timeit(lambda: run(True)), nb + 2 * M * Cn * 2, fl)
        # what the fused forms replace: the stand-alone passes
        t = torch.empty_like(a)
        report(f"      bn_act {H}x{H} C{Cn}", timeit(lambda: kn.bn_act(a, sc, sh, t, relu=True)), 2 * M * Cn * 2)
        report(f"      bn_bwd_apply {H}x{H} C{Cn}", timeit(lambda: kn.bn_bwd_apply(a, c1, sc, sh, sh, t)), 3 * M * Cn * 2)
        del a, y, ident, bits, c1, dc, t
    # the strided-residual class
    for H, C, K in ((56, 256, 128), (28, 512, 256), (14, 1024, 512)):
        if only:
            continue
        N = NIMG
        d = kn.conv_desc(DT, N, H, H, C, K, 1, 1, 1, 0)
        M = N * H * H
        dy = rnd(M, K, scale=0.05)
        w = rnd(K, C, scale=0.05)
        wpk = kn.panel_pack_weights(w, torch.empty_like(w), C, K, 1, C)
        dx = torch.empty(M, C, dtype=DT, device="cuda")
        lo = rnd(N * (H // 2) * (H // 2), C, scale=0.05)
        bits = torch.randint(0, 256, (kn.gate_numel(M, C, DT),), dtype=torch.uint8, device="cuda")

        def run3():
            sums = kn.new_stats(C, 2, "cuda")
            kn.panel_dgrad(d, dy, wpk, dx, resid=lo, resid_stride=2, mask_bits=bits, sums=sums)
        report(f"panel epi3 {H}x{H} dY{K}->dX{C}: bits+sums+lowres residual", timeit(run3),
               (M * K + M * C) * 2 + M * C // 8 + lo.numel() * 2, 2.0 * M * K * C)
        del dy, dx, lo, bits


def case_pool():
    N, H, C = NIMG, 112, 64
    P = H // 2
    c0 = rnd(N * H * H, C)
    sc, sh = torch.ones(C, device="cuda"), torch.zeros(C, device="cuda")
    out = torch.empty(N * P * P, C, dtype=DT, device="cuda")
    am = torch.empty(N * P * P, C, dtype=torch.uint8, device="cuda")
    nb_f = c0.numel() * 2 + out.numel() * 3
    report("pool fwd 112x112x64", timeit(lambda: kn.stem_pool_fwd(c0, sc, sh, out, am, N, H, H, C)), nb_f)
    dp = rnd(N * P * P, C, scale=0.05)
    g0 = torch.empty_like(c0)

    def run():
        sums = kn.new_stats(C, 2, "cuda")
        kn.stem_pool_bwd(dp, am, c0, sc, sh, g0, sums, N, H, H, C)
    nb_b = dp.numel() * 3 + 2 * c0.numel() * 2
    report("pool bwd 112x112x64", timeit(run), nb_b)


def case_fuser():
    """the fuser heads' Linear layers (backbone.py:195-212) at 2 x 256 rows"""
    rows = 512
    for Cin, K in ((18432, 18432), (9216, 9216), (18432, 4608), (4608, 18432), (4608, 4608)):
        d = kn.conv_desc(DT, rows, 1, 1, Cin, K, 1, 1, 1, 0)
        x = rnd(rows, Cin)
        w = rnd(K, Cin, scale=0.02)
        y = torch.empty(rows, K, dtype=DT, device="cuda")
        wb = K * Cin * 2
        flop = 2.0 * rows * Cin * K
        report(f"fuser fwd   {rows}x{Cin} -> {K}", timeit(lambda: kn.conv_fwd(d, x, w, y)), wb + (x.numel() + y.numel()) * 2, flop)
        dy = rnd(rows, K, scale=0.05)
        dx = torch.empty(rows, Cin, dtype=DT, device="cuda")
        report(f"fuser dgrad {rows}x{K} -> {Cin}", timeit(lambda: kn.conv_dgrad(d, dy, w, dx)), wb + (dy.numel() + dx.numel()) * 2, flop)
        dw = torch.zeros(K, Cin, device="cuda")
        report(f"fuser wgrad {rows}: {K}x{Cin}", timeit(lambda: kn.conv_wgrad(d, x, dy, dw)), K * Cin * 8 + (x.numel() + dy.numel()) * 2, flop)
        report(f"fuser wgrad {rows}: {K}x{Cin} STORED", timeit(lambda: kn.conv_wgrad_store(d, x, dy, dw)),
               K * Cin * 4 + (x.numel() + dy.numel()) * 2, flop)
        del x, w, y, dy, dx, dw


def case_dma():
    """plain 1x1 forward / input gradient without epilogue operands (reference points for the classes above)"""
    for H, Cn, K in ((56, 256, 64), (56, 64, 256), (28, 512, 128), (14, 1024, 256), (14, 256, 1024)):
        N = NIMG
        M = N * H * H
        d = kn.conv_desc(DT, N, H, H, Cn, K, 1, 1, 1, 0)
        x = rnd(M, Cn)
        w = rnd(K, Cn, scale=0.05)
        y = torch.empty(M, K, dtype=DT, device="cuda")
        nb = (M * Cn + M * K) * 2
        report(f"dma  {H}x{H} fwd C{Cn}->K{K} (+stats)",
               timeit(lambda: kn.conv_fwd(d, x, w, y, stats=kn.new_stats(K, 2, "cuda"))), nb, 2.0 * M * Cn * K)
        report(f"dma  {H}x{H} dgrad dY{K}->dX{Cn}", timeit(lambda: kn.conv_dgrad(d, y, w, x)), nb, 2.0 * M * Cn * K)
        del x, y


def case_deep():
    """MFMA-bound layers: 3x3 and deep 1x1, forward / input gradient / weight gradient (the A/B shapes of the
    MSFWSI_FETCH_FIRST order)"""
    for H, Cn, K, R in ((14, 256, 256, 3), (7, 512, 512, 3), (28, 128, 128, 3), (14, 1024, 256, 1), (7, 512, 2048, 1)):
        N = NIMG
        M = N * H * H
        d = kn.conv_desc(DT, N, H, H, Cn, K, R, R, 1, R // 2)
        x = rnd(M, Cn)
        w = rnd(K, R * R * Cn, scale=0.05)
        y = torch.empty(M, K, dtype=DT, device="cuda")
        dy = rnd(M, K, scale=0.05)
        dx = torch.empty(M, Cn, dtype=DT, device="cuda")
        dw = torch.zeros(K, R * R * Cn, device="cuda")
        fl = 2.0 * M * Cn * K * R * R
        nb = (M * Cn + M * K) * 2
        report(f"deep {H}x{H} C{Cn}->K{K} {R}x{R} fwd (+stats)",
               timeit(lambda: kn.conv_fwd(d, x, w, y, stats=kn.new_stats(K, 2, "cuda"))), nb, fl)
        report(f"deep {H}x{H} C{Cn}->K{K} {R}x{R} dgrad", timeit(lambda: kn.conv_dgrad(d, dy, w, dx)), nb, fl)
        report(f"deep {H}x{H} C{Cn}->K{K} {R}x{R} wgrad", timeit(lambda: kn.conv_wgrad(d, x, dy, dw)), nb + dw.numel() * 4, fl)
        del x, y, dy, dx, dw


def case_img3():
    """conv2 of layer2 / layer3: the image-stationary kernel (csrc/img3x3.hip) beside the gather kernel the engine runs"""
    for H, Cn in ((14, 256), (28, 128), (56, 64)):
        N = NIMG
        M = N * H * H
        d = kn.conv_desc(DT, N, H, H, Cn, Cn, 3, 3, 1, 1)
        assert kn.img3x3_supported(d)
        x = rnd(M, Cn)
        w = rnd(Cn, 9 * Cn, scale=0.05).view(Cn, 3, 3, Cn)
        wf = kn.img3x3_pack_weights(w, torch.empty_like(w), False)
        wb = kn.img3x3_pack_weights(w, torch.empty_like(w), True)
        y = torch.empty(M, Cn, dtype=DT, device="cuda")
        a = torch.empty(M, Cn, dtype=DT, device="cuda")
        dy = rnd(M, Cn, scale=0.05)
        dx = torch.empty(M, Cn, dtype=DT, device="cuda")
        dc = torch.empty(M, Cn, dtype=DT, device="cuda")
        sc, sh = torch.rand(Cn, device="cuda") + 0.5, torch.randn(Cn, device="cuda") * 0.3
        k1, k2, k3 = torch.rand(Cn, device="cuda"), torch.randn(Cn, device="cuda") * 0.1, torch.randn(Cn, device="cuda") * 0.01
        fl = 2.0 * M * Cn * Cn * 9
        nb = 2 * M * Cn * 2
        tag = f"img3 {H}x{H} C{Cn}"
        if H == 56:  # layer1: the engine runs the weights-stationary kernel (BatchNorm + ReLU already in its staging)
            report(f"{tag} fwd weights-stationary (+stats)", timeit(lambda: kn.conv3x3_fwd(d, x, w, y, stats=kn.new_stats(Cn, 2, "cuda"))), nb, fl)
            report(f"{tag} fwd weights-stationary, BN+ReLU in the staging",
                   timeit(lambda: kn.conv3x3_fwd(d, x, w, y, stats=kn.new_stats(Cn, 2, "cuda"), pro=(sc, sh))), nb, fl)
            report(f"{tag} dgrad weights-stationary + gate",
                   timeit(lambda: kn.conv3x3_dgrad(d, dy, w, dx, mask=(x, sc, sh), sums=kn.new_stats(Cn, 2, "cuda"))), nb + M * Cn * 2, fl)
            report(f"{tag} bn_bwd_apply + dgrad weights-stationary + gate",
                   timeit(lambda: (kn.bn_bwd_apply(dy, y, k1, k2, k3, dc),
                                   kn.conv3x3_dgrad(d, dc, w, dx, mask=(x, sc, sh), sums=kn.new_stats(Cn, 2, "cuda")))),
                   nb * 2 + 2 * M * Cn * 2, fl)
        report(f"{tag} fwd gather (+stats)", timeit(lambda: kn.conv_fwd(d, x, w, y, stats=kn.new_stats(Cn, 2, "cuda"))), nb, fl)
        report(f"{tag} fwd image (+stats)", timeit(lambda: kn.img3x3_fwd(d, x, wf, y, stats=kn.new_stats(Cn, 2, "cuda"))), nb, fl)
        report(f"{tag} bn_act + fwd gather", timeit(lambda: (kn.bn_act(x, sc, sh, a, relu=True),
                                                          kn.conv_fwd(d, a, w, y, stats=kn.new_stats(Cn, 2, "cuda")))), nb * 2, fl)
        report(f"{tag} fwd image, BN+ReLU in the staging",
               timeit(lambda: kn.img3x3_fwd(d, x, wf, y, stats=kn.new_stats(Cn, 2, "cuda"), pro=(sc, sh))), nb, fl)
        report(f"{tag} dgrad gather + gate", timeit(lambda: kn.conv_dgrad(d, dy, w, dx, mask=(x, sc, sh), sums=kn.new_stats(Cn, 2, "cuda"))),
               nb + M * Cn * 2, fl)
        report(f"{tag} dgrad image + gate", timeit(lambda: kn.img3x3_dgrad(d, dy, wb, dx, mask=(x, sc, sh), sums=kn.new_stats(Cn, 2, "cuda"))),
               nb + M * Cn * 2, fl)
        report(f"{tag} bn_bwd_apply + dgrad gather + gate",
               timeit(lambda: (kn.bn_bwd_apply(dy, y, k1, k2, k3, dc),
                               kn.conv_dgrad(d, dc, w, dx, mask=(x, sc, sh), sums=kn.new_stats(Cn, 2, "cuda")))), nb * 2 + 2 * M * Cn * 2, fl)
        report(f"{tag} dgrad image + gate, BN backward in the staging (dc written)",
               timeit(lambda: kn.img3x3_dgrad(d, dy, wb, dx, bnbwd=(y, k1, k2, k3), dc_out=dc, mask=(x, sc, sh),
                                              sums=kn.new_stats(Cn, 2, "cuda"))), nb * 2 + M * Cn * 2, fl)
        del x, y, a, dy, dx, dc


def case_s2pro():
    """the strided 3x3 conv2 of layer2.0 / layer3.0 / layer4.0 (resnet.py:128 with stride 2): BatchNorm + ReLU of the operand
    materialised first (what the engine does) against applied in the gather kernel's register staging"""
    for H, Cn in ((56, 128), (28, 256), (14, 512)):
        N = NIMG
        M = N * H * H
        d = kn.conv_desc(DT, N, H, H, Cn, Cn, 3, 3, 2, 1)
        x = rnd(M, Cn)
        a = torch.empty(M, Cn, dtype=DT, device="cuda")
        w = rnd(Cn, 9 * Cn, scale=0.05).view(Cn, 3, 3, Cn)
        y = torch.empty(N * d.P * d.Q, Cn, dtype=DT, device="cuda")
        dy = rnd(N * d.P * d.Q, Cn, scale=0.05)
        dw = torch.zeros(Cn, 9 * Cn, device="cuda")
        sc, sh = torch.rand(Cn, device="cuda") + 0.5, torch.randn(Cn, device="cuda") * 0.3
        fl = 2.0 * N * d.P * d.Q * Cn * Cn * 9
        nb = (M * Cn + N * d.P * d.Q * Cn) * 2
        tag = f"s2pro {H}x{H} C{Cn}"
        report(f"{tag} fwd, operand materialised", timeit(lambda: kn.conv_fwd(d, x, w, y, stats=kn.new_stats(Cn, 2, "cuda"))), nb, fl)
        report(f"{tag} bn_act + fwd", timeit(lambda: (kn.bn_act(x, sc, sh, a, relu=True),
                                                     kn.conv_fwd(d, a, w, y, stats=kn.new_stats(Cn, 2, "cuda")))), nb + 2 * M * Cn * 2, fl)
        report(f"{tag} fwd, BN+ReLU in the staging", timeit(lambda: kn.conv_fwd(d, x, w, y, stats=kn.new_stats(Cn, 2, "cuda"), pro=(sc, sh))), nb, fl)
        report(f"{tag} bn_act + wgrad", timeit(lambda: (kn.bn_act(x, sc, sh, a, relu=True), kn.conv_wgrad(d, a, dy, dw))),
               nb + 2 * M * Cn * 2, fl)
        report(f"{tag} wgrad, BN+ReLU in the staging", timeit(lambda: kn.conv_wgrad(d, x, dy, dw, pro=(sc, sh))), nb, fl)
        del x, a, y, dy, dw


def case_gramsplit():
    """small-M weight-gradient / Gram launches of the context passes (N = 256) against the cap on the pixel splits (tuning
    key 15): every split ends with an atomic add of its whole tile (fp64 for the Gram)"""
    lib = _lib.load()
    for N, H, Cn, K, gram in ((256, 14, 256, 256, True), (256, 14, 256, 1024, False), (256, 14, 1024, 256, False),
                              (256, 7, 512, 512, True), (256, 7, 512, 2048, False), (256, 28, 128, 512, False)):
        M = N * H * H
        d = kn.conv_desc(DT, N, H, H, Cn, K, 1, 1, 1, 0)
        x = rnd(M, Cn)
        dy = rnd(M, K, scale=0.05)
        dw = torch.zeros(K, Cn, device="cuda")
        A = torch.zeros(Cn, 1, 1, Cn, device="cuda")
        for cap in (0, 128, 64, 32, 16):
            lib.msfwsi_set_tuning(15, cap)
            if gram:
                ms = timeit(lambda: kn.gram(kn.conv_desc(DT, N, H, H, Cn, Cn, 1, 1, 1, 0), x, A), iters=20)
            else:
                ms = timeit(lambda: kn.conv_wgrad(d, x, dy, dw), iters=20)
            report(f"{'gram ' if gram else 'wgrad'} N{N} {H}x{H} C{Cn}->K{Cn if gram else K} max splits {cap or 'none'}", ms,
                   (M * Cn + (0 if gram else M * K)) * 2, 2.0 * M * Cn * (Cn if gram else K))
        lib.msfwsi_set_tuning(15, 0)


def case_s2dgrad():
    """the strided conv2's input gradient (layer2.0 / layer3.0): four parity launches of the gather kernel (+ the stand-alone
    BatchNorm passes around them) against the one-launch image-stationary kernel with both folded in"""
    for P, Cn in ((28, 128), (14, 256)):
        N, H = NIMG, 2 * P
        d = kn.conv_desc(DT, N, H, H, Cn, Cn, 3, 3, 2, 1)
        Mo, Mi = N * P * P, N * H * H
        dy, c2 = rnd(Mo, Cn, scale=0.05), rnd(Mo, Cn)
        dc = torch.empty(Mo, Cn, dtype=DT, device="cuda")
        c1 = rnd(Mi, Cn)
        a1 = torch.empty(Mi, Cn, dtype=DT, device="cuda")
        dx = torch.empty(Mi, Cn, dtype=DT, device="cuda")
        w = rnd(Cn, 9 * Cn, scale=0.05).view(Cn, 3, 3, Cn)
        wpk = kn.img3x3_pack_weights(w, torch.empty_like(w), 2)
        sc, sh = torch.rand(Cn, device="cuda") + 0.5, torch.randn(Cn, device="cuda") * 0.3
        k1, k2, k3 = torch.rand(Cn, device="cuda"), torch.randn(Cn, device="cuda") * 0.1, torch.randn(Cn, device="cuda") * 0.01
        fl = 2.0 * Mo * Cn * Cn * 9
        nb = (Mo + 2 * Mi) * Cn * 2
        tag = f"s2dgrad {P}x{P} -> {H}x{H} C{Cn}"
        report(f"{tag} gather (4 launches) + gate", timeit(lambda: kn.conv_dgrad(d, dy, w, dx, mask=(c1, sc, sh), sums=kn.new_stats(Cn, 2, "cuda"))), nb, fl)
        report(f"{tag} image + gate", timeit(lambda: kn.img3x3_s2_dgrad(d, dy, wpk, dx, mask=(c1, sc, sh), sums=kn.new_stats(Cn, 2, "cuda"))), nb, fl)
        report(f"{tag} bn_bwd_apply + gather + gate + bn_act (a1)",
               timeit(lambda: (kn.bn_bwd_apply(dy, c2, k1, k2, k3, dc), kn.conv_dgrad(d, dc, w, dx, mask=(c1, sc, sh), sums=kn.new_stats(Cn, 2, "cuda")),
                               kn.bn_act(c1, sc, sh, a1, relu=True))), nb + 3 * Mo * Cn * 2 + 2 * Mi * Cn * 2, fl)
        report(f"{tag} image + gate, BN backward in the staging, a1 from the gate",
               timeit(lambda: kn.img3x3_s2_dgrad(d, dy, wpk, dx, bnbwd=(c2, k1, k2, k3), dc_out=dc, mask=(c1, sc, sh),
                                                 sums=kn.new_stats(Cn, 2, "cuda"), act_out=a1)), nb + 2 * Mo * Cn * 2 + Mi * Cn * 2, fl)
        del dy, c2, dc, c1, a1, dx


def case_mwgrad():
    """the folded tail's M = g^T a2 launch (conv3's weight-gradient basis): a2 = relu(bn2(c2)) materialised by a
    bn_act pass first (what the engine does) against normalised in the weight-gradient kernel's register staging"""
    for H, Cw in ((56, 64), (28, 128), (14, 256), (7, 512)):
        N, K = NIMG, 4 * Cw
        M = N * H * H
        d = kn.conv_desc(DT, N, H, H, Cw, K, 1, 1, 1, 0)
        c2 = rnd(M, Cw)
        a2 = torch.empty(M, Cw, dtype=DT, device="cuda")
        g = rnd(M, K, scale=0.05)
        dw = torch.zeros(K, Cw, device="cuda")
        sc, sh = torch.rand(Cw, device="cuda") + 0.5, torch.randn(Cw, device="cuda") * 0.3
        fl = 2.0 * M * Cw * K
        nb = (M * Cw + M * K) * 2
        tag = f"mwgrad {H}x{H} a2[{Cw}] x g[{K}]"
        report(f"{tag} wgrad, operand materialised", timeit(lambda: kn.conv_wgrad(d, a2, g, dw)), nb, fl)
        report(f"{tag} bn_act + wgrad", timeit(lambda: (kn.bn_act(c2, sc, sh, a2, relu=True), kn.conv_wgrad(d, a2, g, dw))), nb + 2 * M * Cw * 2, fl)
        report(f"{tag} wgrad, BN+ReLU in the staging", timeit(lambda: kn.conv_wgrad(d, c2, g, dw, pro=(sc, sh))), nb, fl)
        report(f"{tag} wgrad, BN+ReLU in the staging, a2 written from there",
               timeit(lambda: kn.conv_wgrad_act(d, c2, g, dw, (sc, sh), a2)), nb + M * Cw * 2, fl)
        del c2, a2, g, dw


def case_wgradreg():
    """HBM-bound 1x1 weight gradients at 56 x 56 whose operand is a block input (non-negative): the DMA-staged launch against
    the register-staged one with a unit BatchNorm (relu(1 * x + 0) = x)"""
    for H, Cn, K in ((56, 256, 64), (56, 256, 128), (56, 64, 64), (56, 64, 256), (28, 512, 128)):
        N = NIMG
        M = N * H * H
        d = kn.conv_desc(DT, N, H, H, Cn, K, 1, 1, 1, 0)
        x = rnd(M, Cn).abs_()
        dy = rnd(M, K, scale=0.05)
        dw = torch.zeros(K, Cn, device="cuda")
        one, zero = torch.ones(Cn, device="cuda"), torch.zeros(Cn, device="cuda")
        fl = 2.0 * M * Cn * K
        nb = (M * Cn + M * K) * 2
        tag = f"wgradreg {H}x{H} x[{Cn}] dy[{K}]"
        report(f"{tag} DMA-staged", timeit(lambda: kn.conv_wgrad(d, x, dy, dw)), nb, fl)
        report(f"{tag} register-staged (unit prologue)", timeit(lambda: kn.conv_wgrad(d, x, dy, dw, pro=(one, zero))), nb, fl)
        del x, dy, dw


def case_dgrad2pro():
    """the folded tail's two-source input gradient: second source materialised (a2 read, c2 read as the gate's tensor) against
    msfwsi_conv_dgrad2_pro (c2 read once: source and gate), and the bn_act pass the layer1 form no longer needs"""
    for H, K, Cw in ((56, 256, 64), (28, 512, 128), (14, 1024, 256), (7, 2048, 512)):
        N = NIMG
        M = N * H * H
        d = kn.conv_desc(DT, N, H, H, Cw, K, 1, 1, 1, 0)
        g = rnd(M, K, scale=0.05)
        c2 = rnd(M, Cw)
        a2 = torch.empty_like(c2)
        sc, sh = torch.ones(Cw, device="cuda"), torch.zeros(Cw, device="cuda")
        wcat = rnd(K + Cw, Cw, scale=0.05)
        bias = torch.zeros(Cw, device="cuda")
        da = torch.empty(M, Cw, dtype=DT, device="cuda")
        fl = 2.0 * M * (K + Cw) * Cw
        report(f"dgrad2pro {H}x{H} bn_act (c2 -> a2)", timeit(lambda: kn.bn_act(c2, sc, sh, a2, relu=True)), 2 * M * Cw * 2)
        report(f"dgrad2pro {H}x{H} g[{K}] a2[{Cw}] materialised, gate from c2",
               timeit(lambda: kn.conv_dgrad2(d, g, wcat, da, a2, bias=bias, mask=(c2, sc, sh), sums=kn.new_stats(Cw, 2, "cuda"))),
               (M * K + 3 * M * Cw) * 2, fl)
        report(f"dgrad2pro {H}x{H} g[{K}] c2[{Cw}] normalised in the launch",
               timeit(lambda: kn.conv_dgrad2(d, g, wcat, da, c2, bias=bias, mask=(c2, sc, sh), sums=kn.new_stats(Cw, 2, "cuda"),
                                             src2_pro=(sc, sh))), (M * K + 2 * M * Cw) * 2, fl)
        del g, c2, a2, da


CASES = {"dgrad2pro": case_dgrad2pro, "wgradreg": case_wgradreg, "mwgrad": case_mwgrad, "s2dgrad": case_s2dgrad, "gramsplit": case_gramsplit, "s2pro": case_s2pro, "img3": case_img3, "panel": case_panel, "deep": case_deep, "epi3": case_epi3, "s2": case_s2, "wide": case_wide, "pool": case_pool, "fuser": case_fuser, "dma": case_dma}


def main():
    lib = _lib.load()
    for kv in os.environ.get("KBENCH_TUNE", "").split(","):  # e.g. KBENCH_TUNE="0=1073741824" (msfwsi_set_tuning keys)
        if kv:
            k, v = kv.split("=")
            lib.msfwsi_set_tuning(int(k), int(v))
            print(f"# msfwsi_set_tuning({k}, {v})", flush=True)
    for name in (sys.argv[1:] or list(CASES)):
        CASES[name]()


if __name__ == "__main__":
    main()
