"""GPU: the ResNet-50 trunk AT THE GEOMETRY OF THE HEADLINE (224 x 224 tiles: feature maps of 56 / 28 / 14 / 7 pixels,
/root/reference/src/models/resnet.py:120-140,232-256) against the reference-derived fixture r50enc_b8_s224_div
(tests/golden/make_golden.py `encoder_224`: the REAL reference trunk on eight well-conditioned images, fp64 features and
gradients, the reference's own fp32 and autocast spreads).

Why this file exists: every other reference-pinned ResNet-50 case is 64 x 64 -- maps of 16 / 8 / 4 / 2
pixels, where the image-stationary 3x3 kernels (56 / 28 / 14 only), the strided one-launch gradient, the panel kernels'
production forms, the chunked image-kernel backward and gap_fwd_stride2 never dispatch on their own.  Here the engine runs
with the PRODUCTION DISPATCH FORCED (fill threshold 0, image chunks, panel forward / gradient / Gram on), the launches are
counted so that a silent fallback fails, and the whole trunk meets a reference-derived number.
"""
import numpy as np
import pytest
import torch

from helpers import LOWP_FLOOR, LOWP_TAG, load_golden, lowp_gate, rel, spread_gate
from test_encoder_gpu import _trunk_case, _trunk_oracle, _trunk_product

pytestmark = pytest.mark.gpu
# (the 16-bit tests run under `reproducible_sums` like every whole-model statistical test; the fp32 test must NOT: with the
#  pixel splits capped at one, a weight-gradient tile is ONE fp32 accumulation chain over all 8 x 56 x 56 = 25 088 rows --
#  measured here: product median 1.6e-3 against 2.8e-4 on the default split-K path, whose partial sums cover ~1 000 rows)

CASE = "r50enc_b8_s224_div"


def _production_engine():
    """an Engine with every size gate of the production dispatch lifted: eight images are far below the fill the engine asks
    for before it takes the image-stationary kernels, and below the 1 GiB from which the backward runs in image chunks"""
    from msf_wsi_amd.engine import Engine

    eng = Engine()
    eng.img3x3_min_fill = 0.0
    eng.img3x3_chunk_bytes = 1      # -> chunks of one image (Engine._img3_bwd_chunks)
    assert eng.img3x3 and eng.img3x3_layer1 and eng.img3x3_s2 and eng.panel_fwd and eng.panel_dgrad and eng.panel_gram
    assert eng.gap_stride_fused and eng.fold_bn3 and eng.fold_bn3_fwd and eng.fold_ds_strided
    return eng


def _count_launches(monkeypatch, fault=None):
    """wraps the kernels.py entry points of the production kernels with counters; `fault(name, args, kwargs)` may alter the
    arguments of a launch (fault injection)"""
    from msf_wsi_amd import kernels as kn

    calls = {}

    def wrap(name):
        real = getattr(kn, name)

        def counted(*a, **k):
            if fault is not None:
                a, k = fault(name, a, k)
            ok = real(*a, **k)
            if ok is not False:     # (False: the kernel declined the shape and the engine fell back)
                key = name + ("_pro" if name == "conv_dgrad2" and k.get("src2_pro") is not None else "")
                calls[key] = calls.get(key, 0) + 1
            return ok

        monkeypatch.setattr(kn, name, counted)

    for name in ("img3x3_fwd", "img3x3_dgrad", "img3x3_s2_dgrad", "panel_fwd_post", "panel_dgrad", "panel_gram",
                 "gap_fwd_stride2", "conv3x3_fwd", "conv3x3_dgrad", "conv_wgrad_act", "stem_wgrad_bnbwd", "conv_dgrad2"):
        wrap(name)
    return calls


def test_resnet50_trunk_224_fp32(hip_lib):
    """fp32 (the exact-fp32 MFMA gather kernels; the 16-bit-only stationary kernels do not apply): features 1e-3, every
    gradient tensor against the fp64 oracle of this machine, itself pinned to the reference's fp64 loss by the fixture.
    At 224 x 224 a trunk has 12 x the ReLU gates of the 64 x 64 case: the REFERENCE's own fp32 run of the fixture's seed sits
    9e-4 (median) from its fp64 run -- one gate flip near the top moves everything below it.  Same rule as the 64 x 64 case
    (test_encoder_gpu.test_resnet50_trunk_well_conditioned_fp32): three input seeds, per seed the flip-tolerant gate, per
    tensor the SMALLEST of the three distances at the level of the reference's own fp32 runs treated the same way."""
    vec, man = load_golden(CASE)
    assert man["size"] == 224 and man["B"] == 8 and man["arch"] == "resnet50"
    names = man["param_keys"]
    per_seed, per_seed_ref, boxes = [], [], []
    for seed in (man["data_seed"], 1, 2):
        enc, sd0, x, Rs = _trunk_case(man, seed)
        f64, g64 = _trunk_oracle(sd0, x, Rs, want_loss=float(vec["loss"][0]) if seed == man["data_seed"] else None)
        _, g32 = _trunk_oracle({k: v.clone() for k, v in sd0.items()}, x, Rs, torch.float32)
        feats, grads = _trunk_product(enc, x, Rs, torch.float32)
        if seed == man["data_seed"]:   # the committed reference features themselves (fp32 copies of the fp64 run)
            for s, f in enumerate(feats):
                assert rel(f, vec[f"feat/{s}"]) < 1e-3, (s, rel(f, vec[f"feat/{s}"]))
        fr = np.array([rel(f.float(), r) for f, r in zip(feats, f64)])
        assert fr.max() < 1e-3, fr
        rels = np.array([rel(grads[k], g64[k]) for k in names])
        box = np.array([rel(g32[k], g64[k]) for k in names])
        spreads = [box] + ([vec["spread_grad"]] if seed == man["data_seed"] else [])
        # (size bound of the flip rule: a gate flip moves the tensors below it by 1e-3 .. 1e-2 whichever seed it hits, and
        #  whether the REFERENCE's fp32 run of a particular seed shows one is luck -- measured: seed 2's shows none (max 4e-4)
        #  where the product's has one at layer4.0.bn1 (2.3e-3); the reference's runs of the other seeds are samples of the
        #  same process: `envelope`)
        spread_gate(rels, names, spreads, f"resnet50 trunk 224x224 (seed {seed}), fp32 gradients", count_rule=False,
                    envelope=[vec["spread_grad"]] + boxes)
        boxes.append(box)
        per_seed.append(rels)
        per_seed_ref.append(np.min(np.stack(spreads), axis=0))
        del enc, feats, grads
        torch.cuda.empty_cache()
    best, best_ref = np.min(np.stack(per_seed), axis=0), np.min(np.stack(per_seed_ref), axis=0)
    print(f"[trunk 224 fp32] per-tensor minimum over 3 seeds: product median {np.median(best):.2e} p90 "
          f"{np.quantile(best, .9):.2e} max {best.max():.2e}; the reference's own fp32 runs: median {np.median(best_ref):.2e} "
          f"p90 {np.quantile(best_ref, .9):.2e}")
    assert np.median(best) <= 5.0 * max(float(np.median(best_ref)), 1e-6), float(np.median(best))
    assert np.quantile(best, 0.9) <= max(1e-3, 5.0 * float(np.quantile(best_ref, 0.9)))


def _run_lowp(man, vec, dtype, monkeypatch, fault=None):
    calls = _count_launches(monkeypatch, fault)
    enc, sd0, x, Rs = _trunk_case(man)
    enc._engine = _production_engine()
    f64, g64 = _trunk_oracle(sd0, x, Rs, want_loss=float(vec["loss"][0]))
    feats, grads = _trunk_product(enc, x, Rs, dtype)
    names = man["param_keys"]
    rels = np.array([rel(grads[k], g64[k]) for k in names])
    fr = np.array([rel(f.float(), r) for f, r in zip(feats, f64)])
    return calls, names, rels, fr


def _gate_lowp(vec, dtype, names, rels, fr, what):
    tag = LOWP_TAG[dtype]
    print(f"[{what}] features rel {fr}, reference under autocast {vec[f'spread_feat_{tag}']}")
    assert (fr <= np.maximum(LOWP_FLOOR[dtype], 2.0 * vec[f"spread_feat_{tag}"])).all(), fr
    lowp_gate(rels, names, vec[f"spread_grad_{tag}"], LOWP_FLOOR[dtype], what)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16], ids=["bf16", "fp16"])
def test_resnet50_trunk_224_production_dispatch(hip_lib, reproducible_sums, monkeypatch, dtype):
    """16-bit storage with the production dispatch forced: features and every gradient tensor within 2 x the distance of
    the REFERENCE UNDER AUTOCAST from its fp64 run (fixture spread_*_bf16 / _fp16), and the launches COUNTED:
    conv2 of the stride-1 Bottlenecks of layer2 / layer3 forward on the image kernel (3 + 5), every conv2 gradient of
    layer1-3's stride-1 blocks on it with bn2's backward folded in (2 + 3 + 5 blocks, in one-image chunks where the
    by-product activation exists), the strided gradients of layer2.0 / layer3.0 in one launch per chunk, the panel kernel
    for the fused tails and conv1's gradients, the fused Gram pass, gap_fwd_stride2 for the three stage outputs."""
    vec, man = load_golden(CASE)
    if f"spread_grad_{LOWP_TAG[dtype]}" not in vec:
        pytest.skip("the fixture carries no reference-under-autocast run of this dtype")
    calls, names, rels, fr = _run_lowp(man, vec, dtype, monkeypatch)
    print(f"[trunk 224 {LOWP_TAG[dtype]}] launches: {calls}")
    _gate_lowp(vec, dtype, names, rels, fr, f"resnet50 trunk 224x224 {LOWP_TAG[dtype]}, production dispatch: gradients")
    B = man["B"]
    assert calls.get("img3x3_fwd", 0) == 3 + 5, calls
    # layer1: its 3 blocks (conv2 has stride 1 in all of them), whole batch per launch (no by-product activation -> no
    # chunks); layer2 / layer3: 3 + 5 stride-1 blocks in chunks of one image each
    assert calls.get("img3x3_dgrad", 0) == 3 + (3 + 5) * B, calls
    assert calls.get("img3x3_s2_dgrad", 0) == 2 * B, calls
    assert calls.get("gap_fwd_stride2", 0) == 3, calls
    # fused tails on the panel kernel: every Bottleneck without a downsample branch from 128 channels up (layer2-4: 3 + 5 + 2)
    # plus layer1's two at 64 channels where k >= panel_fwd_min_k allows; conv1's input gradients: all 16 blocks but the first
    assert calls.get("panel_fwd_post", 0) == 12 and calls.get("panel_dgrad", 0) == 15, calls
    # the folded tails' two-source input gradients: layer1's three normalise conv2's raw output inside the launch (a2 is
    # never stored: no by-product weight-gradient launch either), the other thirteen read the materialised a2
    assert calls.get("panel_gram", 0) == 5 and calls.get("conv_dgrad2_pro", 0) == 3 and calls.get("conv_dgrad2", 0) == 13 + 4, calls  # (+ the four folded downsample branches)
    assert calls.get("conv_wgrad_act", 0) == 0, calls
    assert calls.get("conv3x3_fwd", 0) == 3, calls   # layer1's conv2 forward stays weights-stationary


@pytest.mark.parametrize("dtype", [torch.float16], ids=["fp16"])
def test_fault_in_image_kernel_bn_backward_is_caught(hip_lib, reproducible_sums, monkeypatch, dtype):
    """falsifiability at this geometry: the k2 coefficient of bn2's backward (dc = k1 g + k2 c + k3), as the image-stationary
    gradient kernel forms it in its staging, multiplied by 1.5 in every launch -- the whole-trunk gate must turn red.
    (fp16: the reference-under-autocast yardstick is 8 x tighter than bf16's, where one wrong coefficient of a residual
    branch with a trained-like gain of 0.1 stays inside the reference's own 12 % noise; the per-block tests,
    tests/test_blocks_lowp_gpu.py, see x 1.02.)"""
    vec, man = load_golden(CASE)
    if f"spread_grad_{LOWP_TAG[dtype]}" not in vec:
        pytest.skip("the fixture carries no reference-under-autocast run of this dtype")
    hits = []

    def fault(name, a, k):
        if name in ("img3x3_dgrad", "img3x3_s2_dgrad") and k.get("bnbwd") is not None:
            c, k1, k2, k3 = k["bnbwd"]
            k = dict(k, bnbwd=(c, k1, k2 * 1.5, k3))
            hits.append(name)
        return a, k

    calls, names, rels, fr = _run_lowp(man, vec, dtype, monkeypatch, fault)
    assert len(hits) >= 10, "the fault was never injected"
    with pytest.raises(AssertionError):
        _gate_lowp(vec, dtype, names, rels, fr, "fault injection (k2 x 1.5 in the image kernel's BatchNorm backward)")


def test_resnet50_trunk_224_full_size_replicated(hip_lib, reproducible_sums, monkeypatch):
    """The trunk AT THE FULL SIZE of BASELINE config 2's target pass -- 4 096 images of 224 x 224, bf16 -- on the engine's
    NATURAL dispatch (nothing forced: at this size the image-stationary kernels fill the chip on their own, the by-product
    activations exceed 1 GiB so the backward runs in image chunks, the 256 x 128 tiles and the 256 x 256 weight-gradient tile
    are chosen by the launch geometry), tied to the reference fixture by a size-independent property: the batch is the
    fixture's eight images REPEATED 512 times.  Train-mode BatchNorm over 512 identical copies of a batch has the statistics
    of the batch, so every copy's features are the eight-image features and every parameter gradient of
    L = sum_s <features_s, R_s (repeated)> is 512 x the eight-image gradient -- exactly, in exact arithmetic.  Gates: the
    features of the first, a middle and the last copy and the gradients / 512 against the fp64 oracle of the eight images
    (pinned to the reference's fp64 loss), with the same reference-under-autocast yardstick as the eight-image test."""
    free, total = torch.cuda.mem_get_info()
    if total < 250 * 2 ** 30:
        pytest.skip("needs the 288 GB of an MI355X")
    torch.cuda.empty_cache()
    vec, man = load_golden(CASE)
    calls = _count_launches(monkeypatch)
    enc, sd0, x, Rs = _trunk_case(man)
    f64, g64 = _trunk_oracle(sd0, x, Rs, want_loss=float(vec["loss"][0]))
    REP, B = 512, man["B"]
    enc = enc.cuda().train()
    xr = x.cuda().repeat(REP, 1, 1, 1)
    Rr = [r.cuda().repeat(REP, 1) for r in Rs]
    with torch.autocast("cuda", dtype=torch.bfloat16):
        feats = enc(xr)
    loss = sum((f.float() * r).sum() for f, r in zip(feats, Rr))
    loss.backward()
    torch.cuda.synchronize()
    print(f"[trunk 224 x {REP * B} images bf16] launches: {calls}; peak memory {torch.cuda.max_memory_allocated() / 2 ** 30:.1f} GiB")
    # the natural dispatch took the production kernels (whole-batch launches; the image-kernel backward in chunks)
    assert calls.get("img3x3_fwd", 0) == 8 and calls.get("img3x3_s2_dgrad", 0) >= 2 and calls.get("img3x3_dgrad", 0) >= 11, calls
    assert calls.get("panel_fwd_post", 0) == 12 and calls.get("panel_dgrad", 0) == 15 and calls.get("panel_gram", 0) == 5, calls
    assert calls.get("stem_wgrad_bnbwd", 0) == 1 and calls.get("gap_fwd_stride2", 0) == 3, calls
    tag = "bf16"
    for copy in (0, REP // 2, REP - 1):
        fr = np.array([rel(f[copy * B:(copy + 1) * B].float(), r) for f, r in zip(feats, f64)])
        assert (fr <= np.maximum(LOWP_FLOOR[torch.bfloat16], 2.0 * vec[f"spread_feat_{tag}"])).all(), (copy, fr)
    # every copy carries the same features up to the rounding of sums taken in another order
    spread = max(rel(f[:B].float(), f[(REP - 1) * B:].float()) for f in feats)
    assert spread <= 2 * LOWP_FLOOR[torch.bfloat16], spread
    named = dict(enc.named_parameters())
    names = man["param_keys"]
    rels = np.array([rel(named[k].grad.double().cpu() / REP, g64[k]) for k in names])
    lowp_gate(rels, names, vec[f"spread_grad_{tag}"], LOWP_FLOOR[torch.bfloat16],
              f"resnet50 trunk 224x224, {REP * B} images (the fixture's eight x {REP}), bf16, natural dispatch: gradients / {REP}")
    del feats, xr, Rr, enc
    torch.cuda.empty_cache()
