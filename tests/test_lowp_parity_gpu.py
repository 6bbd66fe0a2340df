"""16-bit parity on a real MI355X (the dtype BASELINE config 2 -- the headline -- is quoted in).

The reference trains under autocast by default (tools/ssl_train.py:96-100,441: fp16 with --amp, bf16 with --bf16) and
SURVEY.md 8(d) gates 16-bit runs on the loss curve.  The yardstick of every test here is the REFERENCE ITSELF UNDER
AUTOCAST, recorded in the fixtures by tests/golden/make_golden.py from the real reference on well-conditioned inputs
(oracle.diverse_batch): per gradient tensor / output tensor / loss term the distance of the reference's 16-bit run from
its own fp64 run (`spread_*_bf16`, `spread_*_fp16`), and the reference's 30-step loss curves in fp64, fp32, bf16 and fp16.
The product's 16-bit run is held, per tensor, to max(2 ulp of the storage type, 2 x that distance) against the fp64
oracle of this machine -- exactly as the fp32 run is held to max(1e-3, 2 x the fp32<->fp64 spread).  No criterion
compares the product with itself, and a deliberately wrong BatchNorm scale in one epilogue turns the gates red
(test_wrong_epilogue_scale_is_caught)."""
import numpy as np
import pytest
import torch

from helpers import (LOWP_FLOOR, LOWP_TAG, LR, build_case, case_batch, flat_outputs, load_golden, lowp_gate,
                     oracle_case, reference_loop_loss, rel)

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("reproducible_sums")]
FP16_LOSS_SCALE = 1024.0  # the fixed power of two the fixture's fp16 reference run used


def term_scale(spread_terms):
    """per loss term: max(its own reference distance, the RMS over the 4 scales of its group)"""
    sp = np.asarray(spread_terms, dtype=np.float64).reshape(3, 4)
    return np.maximum(sp, np.sqrt((sp ** 2).mean(axis=1, keepdims=True))).reshape(-1)


def lowp_step(case, dtype, mutate=None, model=None):
    """the reference loop's statements (tools/ssl_train.py:441-472) under torch.autocast("cuda", dtype) on the product;
    returns the gate inputs.  mutate(model): test hook applied before the step (fault injection).  model: a product model
    of the case built by the caller (repeated runs of one step: gradients are cleared here)"""
    vec, man = load_golden(case)
    oc = oracle_case(case)
    if model is None:
        model = build_case(man).cuda().train()
    for p in model.parameters():
        p.grad = None
    (c1, c2), (t1, t2), idx = case_batch(man)
    scale = FP16_LOSS_SCALE if dtype == torch.float16 else 1.0
    if mutate is not None:
        mutate(model)
    with torch.autocast("cuda", dtype=dtype):
        outs = model((c1.cuda(), t1.cuda()), (c2.cuda(), t2.cuda()), idx)
        loss, terms = reference_loop_loss(outs)
    (loss * scale).backward()
    torch.cuda.synchronize()
    named = list(model.named_parameters())
    assert [n for n, _ in named] == oc["names"]
    grads = [(n, p.grad.double().cpu() / scale) for n, p in named]
    assert all(bool(torch.isfinite(g).all()) for _, g in grads)
    return vec, man, oc, outs, terms.cpu().double(), grads


def gate_lowp_step(case, dtype, what, mutate=None, model=None):
    tag, floor = LOWP_TAG[dtype], LOWP_FLOOR[dtype]
    vec, man, oc, outs, terms, grads = lowp_step(case, dtype, mutate, model)
    # ---- outputs p / z: per tensor against the fp64 oracle, allowance from the reference under autocast
    fo, fr = flat_outputs(outs), flat_outputs(oc["outs64"])
    keys = list(fo)
    assert all(fo[k].dtype == dtype for k in keys)
    lowp_gate([rel(fo[k].float(), fr[k]) for k in keys], ["/".join(map(str, k)) for k in keys], vec[f"spread_out_{tag}"],
              floor, f"{what}: outputs", max_violations=0.1)
    # ---- the 12 loss terms.  One reference sample's |d| of a single term can be near zero by chance (and the product's
    # is another draw of the same noise): the allowance of a term is 2 x the larger of its own reference distance and the
    # RMS distance over the 4 terms of its group, and never below 3 x the RMS over all 12 -- still a few 1e-4 of a cosine
    # in [-1, 1], two orders below what a wrong kernel moves it by (test_wrong_epilogue_scale_is_caught)
    d = (terms - oc["terms64"]).abs().numpy()
    sp = np.asarray(vec[f"spread_terms_{tag}"], dtype=np.float64)
    allow = np.maximum(1e-3 * np.maximum(oc["terms64"].abs().numpy(), 1e-2),
                       np.maximum(2.0 * term_scale(sp), 3.0 * float(np.sqrt((sp ** 2).mean()))))
    print(f"[{what}] loss terms: max |d| {d.max():.2e}, allowance min {allow.min():.2e} max {allow.max():.2e}; "
          f"{int((d > allow).sum())}/12 beyond")
    assert (d <= allow).all(), (d, allow)
    # ---- gradients
    lowp_gate([rel(g, oc["grads64"][n]) for n, g in grads], [n for n, _ in grads], vec[f"spread_grad_{tag}"], floor,
              f"{what}: gradients")


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16], ids=["bf16", "fp16"])
def test_lowp_step_within_reference_autocast_spread_r18(hip_lib, dtype):
    """ResNet-18 dual-stream, 16 tile pairs of 64x64, well-conditioned inputs: outputs, loss terms and every gradient
    of the product's 16-bit step against the fp64 oracle, per tensor within 2 x the reference-under-autocast distance"""
    gate_lowp_step("r18_b16_s64_div", dtype, f"r18_b16_s64_div {LOWP_TAG[dtype]}")


def test_lowp_step_within_reference_autocast_spread_r50(hip_lib):
    """the ResNet-50-derived model the bench runs (folded Bottleneck tails, two-source launches, 18432-wide fuser
    GEMMs), bf16, on well-conditioned inputs"""
    gate_lowp_step("r50_b8_s64_div", torch.bfloat16, "r50_b8_s64_div bf16")


def test_stationary_kernels_within_reference_autocast_spread(hip_lib):
    """the shape-specialised persistent kernels (weights-stationary 3x3, output-stationary weight gradient, stem) are
    2-byte only and size-gated: force them onto the small case and hold the bf16 step to the SAME reference yardstick
    as the gather kernels (no product-vs-product criterion)"""
    from msf_wsi_amd import kernels as kn

    d1 = kn.conv_desc(torch.bfloat16, 16 * 16, 16, 16, 64, 64, 3, 3, 1, 1)  # layer1 of the target pass at 64x64
    try:
        for key in (9, 10, 12):
            hip_lib.msfwsi_set_tuning(key, 1)
        hip_lib.msfwsi_set_tuning(11, 0)   # no size thresholds: the small batch takes the persistent kernels
        hip_lib.msfwsi_set_tuning(13, 0)
        assert kn.conv3x3_stationary(d1) and kn.conv_wgrad_stationary(d1)
        gate_lowp_step("r18_b16_s64_div", torch.bfloat16, "stationary kernels, bf16")
    finally:
        hip_lib.msfwsi_set_tuning(11, 32 * 256 * 256)
        hip_lib.msfwsi_set_tuning(13, 32 * 512 * 256)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16], ids=["bf16", "fp16"])
def test_fused_step_lowp_first_step(hip_lib, dtype):
    """the FUSED step (msf_wsi_amd.train.PretrainStep: HIP cosine loss, flat gradient buffers, device GradScaler) in
    16-bit storage: loss and every gradient of its first step against the fp64 oracle, same yardstick as above"""
    from msf_wsi_amd.train import PretrainStep

    case = "r18_b16_s64_div"
    vec, man = load_golden(case)
    oc = oracle_case(case)
    tag, floor = LOWP_TAG[dtype], LOWP_FLOOR[dtype]
    model = build_case(man).cuda().train()
    ts = PretrainStep(model, lr=LR, global_batch=man["B"], dtype=dtype,
                      init_scale=65536.0 if dtype == torch.bfloat16 else FP16_LOSS_SCALE)
    (c1, c2), (t1, t2), idx = case_batch(man)
    ts.flats.zero_grads()
    outs, rec, dps = ts.forward_loss(((c1.cuda(), c2.cuda()), (t1.cuda(), t2.cuda()), idx), want_grad=True)
    loss = float(ts.loss_accum)
    ts.engine.model_backward(model, rec, dps, ts.grads, dtype)
    torch.cuda.synchronize()
    w = np.tile(np.array([0.1, 0.4, 0.7, 1.0]), 3)
    allow = float((w * np.maximum(1e-3 * np.maximum(oc["terms64"].abs().numpy(), 1e-2),
                                  2.0 * term_scale(vec[f"spread_terms_{tag}"]))).sum())
    print(f"[fused {tag}] loss {loss:.6f} fp64 oracle {oc['loss64']:.6f} allowance {allow:.2e}")
    assert abs(loss - oc["loss64"]) <= allow
    named = list(model.named_parameters())
    scale = float(ts.scale.item())
    rels = [rel(ts.grads.logical(p).double().cpu() / scale, oc["grads64"][n]) for n, p in named]
    lowp_gate(rels, [n for n, _ in named], vec[f"spread_grad_{tag}"], floor, f"fused step {tag}: gradients")


@pytest.mark.parametrize("dtype,factor", [(torch.bfloat16, 1.25), (torch.float32, 1.01)], ids=["bf16", "fp32"])
def test_wrong_epilogue_scale_is_caught(hip_lib, monkeypatch, dtype, factor):
    """falsifiability: the BatchNorm scale of ONE residual-block epilogue multiplied by `factor` (a wrong post_scale)
    must turn the gates red -- 16-bit: the reference-autocast yardstick; fp32: the 1e-3 output gate"""
    from msf_wsi_amd import kernels as kn

    real = kn.bn_act
    hits = []

    def wrong(c, scale, shift, out, ident=None, **kw):
        if ident is not None and not hits and c.shape[-1] == 128:  # first block end of layer2, first encoder pass
            hits.append(1)
            scale = scale * factor
        return real(c, scale, shift, out, ident=ident, **kw)

    monkeypatch.setattr(kn, "bn_act", wrong)
    if dtype == torch.float32:
        from test_parity_gpu import run_reference_loop_case

        with pytest.raises(AssertionError):
            run_reference_loop_case("r18_b16_s64_div")
    else:
        with pytest.raises(AssertionError):
            gate_lowp_step("r18_b16_s64_div", dtype, "fault injection")
    assert hits, "the fault was never injected"


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16], ids=["bf16", "fp16"])
def test_loss_curve_tracks_reference(hip_lib, dtype):
    """SURVEY.md 8(d) "bf16 runs: loss-curve parity": 30 steps of the fused step (HIP loss + backward + Adam +
    GradScaler) in 16-bit storage on a fresh well-conditioned batch per step, against the REAL reference's curves
    (fixture r18_b16_s64_curve: fp64, fp32 and autocast runs of the reference, several samples each).

    The trajectory is chaotic -- Adam's early steps are sign-like, so two fp32 implementations that agree to 1e-6 on a
    step are 1e-3 apart after 5 steps and 5e-2 after 25; the reference-under-autocast(bf16) ends anywhere between
    -0.74 and -0.81 (mean of the last ten losses; samples of the fixture and the oracle re-run on the GPU box's host,
    tools/curve_diag.py) -- so the yardstick is the SPREAD OF THE REFERENCE'S OWN SAMPLES around its fp64 curve:
      per step   |product - fp64| <= max(2e-3, 2 x the largest |sample - fp64| any reference sample has shown up to
                 that step)   (samples: fp32 runs, the oracle's fp32 run, the autocast runs of this dtype); the product
                 is one more draw of the same chaos (its weight gradients are summed with atomics: two runs of the
                 product itself part ways the same way), so up to 3 of the 30 steps may reach 4 x;
      as a whole the mean of the last ten losses within 2 x the largest deviation of a reference sample's mean from
                 the fp64 mean."""
    from msf_wsi_amd.train import PretrainStep
    from oracle import msfwsi_oracle as orc

    vec, man = load_golden("r18_b16_s64_curve")
    tag = LOWP_TAG[dtype]
    B, size = man["B"], man["size"]
    steps = len(vec[f"loss_{tag}"])  # bf16: 30; fp16: the first few (fp16 has no fast CPU path for the reference run)
    ref64 = vec["loss_fp64"][:steps]
    samples = {k[5:]: v[:steps] for k, v in vec.items() if k.startswith("loss_") and k != "loss_fp64"
               and (k.startswith("loss_fp32") or k.startswith(f"loss_{tag}")) and len(v) >= steps}
    samples["oracle_fp32"] = vec["loss_fp32"][:steps] + vec["oracle_fp32_dev"][:steps]  # |dev| stored: one side suffices
    assert len(samples) >= (5 if steps >= 30 else 3), sorted(samples)
    if dtype == torch.bfloat16:
        # ... and the ORACLE under torch.autocast("cpu", bfloat16) on THIS machine: bf16 kernels differ by CPU (the build
        # container's samples end at -0.81 .. -0.84, an EPYC 9575F with native AVX512-BF16 at -0.74), and the oracle's
        # autocast mode is pinned to the reference's (make_golden.py: same forward bit for bit in one process)
        # (a background CPU worker computes it while the GPU runs the other tests: tests/oracle_jobs.py)
        import oracle_jobs

        here = oracle_jobs.get("curve_oracle", "r18_b16_s64_curve", "bf16")["losses"]
        samples["oracle_bf16_this_machine"] = np.array(here)
    env = np.max(np.stack([np.maximum.accumulate(np.abs(v - ref64)) for v in samples.values()]), axis=0)
    allow = np.maximum(2e-3, 2.0 * env)
    model = build_case(man).cuda().train()
    ts = PretrainStep(model, lr=LR, global_batch=B, dtype=dtype,
                      init_scale=65536.0 if dtype == torch.bfloat16 else man["fp16_loss_scale"])
    losses = []
    # (reproducible_sums: one workgroup per weight-gradient tile -- 30 chaotic steps would turn the default's 4e-7 atomics
    #  jitter into a visibly different curve from run to run; with it this test has ONE outcome per build)
    for t in range(steps):
        (c1, c2), (t1, t2), idx = orc.diverse_batch(B, size, 16, man["curve_seed0"] + t)
        losses.append(ts.step(((c1.cuda(), c2.cuda()), (t1.cuda(), t2.cuda()), idx)))
    losses = torch.stack(losses).cpu().numpy().ravel()
    d = np.abs(losses - ref64)
    print(f"[{tag} curve] product " + " ".join(f"{v:.4f}" for v in losses))
    print(f"[{tag} curve] ref fp64" + " ".join(f"{v:.4f}" for v in ref64))
    print(f"[{tag} curve] |d|     " + " ".join(f"{v:.4f}" for v in d))
    print(f"[{tag} curve] allow   " + " ".join(f"{v:.4f}" for v in allow) + f"   ({len(samples)} reference samples)")
    assert np.isfinite(losses).all()
    over = d > allow
    assert over.sum() <= (3 if steps >= 30 else 0) and (d <= 2.0 * allow).all(), (
        int(over.sum()), int(np.argmax(d / allow)), float((d / allow).max()))
    if steps >= 30:
        assert (ref64[-1] - ref64[0]) < -0.5, "the fixture's curve must move for this test to mean anything"
        tail = lambda c: float(np.mean(c[-10:]))
        dev_ref = max(abs(tail(v) - tail(ref64)) for v in samples.values())
        print(f"[{tag} curve] mean of the last 10 steps: product {tail(losses):.4f}, reference fp64 {tail(ref64):.4f}, "
              f"reference samples {min(tail(v) for v in samples.values()):.4f} .. {max(tail(v) for v in samples.values()):.4f}")
        assert abs(tail(losses) - tail(ref64)) <= 2.0 * dev_ref
    assert ts.found_inf.item() == 0 and ts.t == steps  # no step was skipped by the GradScaler
    torch.cuda.synchronize()  # the inter_ group's last Adam pass runs on the optimizer stream
    for gi in range(3):  # the 16-bit compute copies follow the fp32 master weights
        assert torch.equal(ts.flats.w16[gi].float(), ts.flats.w[gi].to(dtype).float())
