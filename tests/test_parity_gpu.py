"""Whole-step parity on a real MI355X: the product (HIP engine behind the reference's model API, driven by
the reference's own loop statements) against the CPU oracle on identical seeded weights and inputs, and
against the committed golden vectors produced from the real reference.

Tolerances (north star: 1e-3 relative, fp32):
  outputs p/z    : rel-L2 <= 1e-3 vs the fp64 oracle and vs the golden tensors (measured ~1e-5 .. 9e-5)
  loss terms     : |d| <= 1e-3 * max(|ref|, 1e-2)
  gradients      : per tensor rel-L2 vs the fp64 oracle <= max(1e-3, 2 x the reference's own fp32<->fp64 spread of that
                   tensor) -- helpers.spread_gate; the spread comes from the fixture (`spread_grad`, the REAL reference
                   in fp32 and fp64, tests/golden/make_golden.py) and from the oracle's fp32/fp64 runs on this machine.
  updated weights: the same rule on w1 after one Adam step with the fixture's `spread_step`.
The reference's own fp32 run is NOT within 1e-3 of its fp64 run on many tensors (r18_b8_s64: 36 gradient tensors up to
3.1e-3; r18_b8_s224: 129 up to 1.8e-2; one BatchNorm bias moves by 25 % after the sign-like first Adam step): a ReLU
gate flips when a pre-activation lies inside the forward rounding noise, and the jump lands on different tensors in
different runs -- hence rule 2 of spread_gate.  Each kernel alone is held to 2e-5 in test_kernels_gpu.py.
"""
import numpy as np
import pytest
import torch

from helpers import (LR, WEIGHTS, build_case, build_product, case_batch, flat_outputs, grad_rels, load_golden, other_spreads,
                     reference_loop_loss, rel, spread_gate, updated_weights_gate)

pytestmark = pytest.mark.gpu


def run_reference_loop_case(case, check_golden_outputs=True):
    """the reference loop's own statements (tools/ssl_train.py:442-474, fp32) on the product, checked against the
    fp64 / fp32 oracle of this machine and the golden vectors of `case`"""
    from oracle import msfwsi_oracle as orc

    vec, man = load_golden(case)
    B, size = man["B"], man["size"]
    diverse = man.get("input_kind", "normal") == "diverse"
    model = build_case(man)
    sd0 = {k: v.detach().clone() for k, v in model.state_dict().items()}
    # seeded construction reproduces the reference's initialisation (pinned by checksums)
    assert [k for k in sd0] == [k for k, _, _ in man["keys"]]
    got_sum = np.array([float(v.double().sum()) for v in sd0.values()])
    assert np.allclose(got_sum, vec["init_sum"], rtol=1e-9, atol=1e-9)

    batch = case_batch(man)
    do_adam = bool(man["adam"])
    # one fp64 and one fp32 oracle step of this case on this machine, shared with the other tests of the session
    # (helpers.oracle_case; it asserts the oracle's fp64 loss terms against the fixture of the real reference)
    from helpers import oracle_case

    oc = oracle_case(case)
    loss64, terms64, outs64, grads64, sd64 = oc["loss64"], oc["terms64"], oc["outs64"], oc["grads64"], oc["sd64"]
    assert abs(float(loss64) - float(vec["loss"][0])) < 1e-7

    model = model.cuda().train()
    (c1, c2), (t1, t2), idx = batch
    lr = LR * (B ** 0.5) / (32 ** 0.5)
    named = list(model.named_parameters())
    groups = [[p for n, p in named if n.startswith(pre)] for pre in ("context_", "target_", "inter_")]
    opt = torch.optim.Adam([{"params": g, "lr": lr} for g in groups], lr=lr)
    # idx stays on the CPU as in the reference loop
    outs = model((c1.cuda(), t1.cuda()), (c2.cuda(), t2.cuda()), idx)
    loss, terms = reference_loop_loss(outs)
    opt.zero_grad()
    loss.backward()
    torch.cuda.synchronize()

    # ---- outputs: vs fp64 oracle and vs the golden tensors of the real reference
    fo, fr = flat_outputs(outs), flat_outputs(outs64)
    worst = max(rel(fo[k], fr[k]) for k in fo)
    ref_out = float(np.max(vec["spread_out"])) if "spread_out" in vec else float("nan")
    print(f"[{case}] outputs p/z vs fp64 oracle: worst rel-L2 {worst:.2e} (the reference's own fp32 run: {ref_out:.2e})")
    assert worst < 1e-3, worst
    for (g, kind, s), t in fo.items():
        assert t.requires_grad == (kind in ("p1", "p2"))
        if check_golden_outputs:
            rows = t if g != "target" else t[:: max(1, t.shape[0] // 8)][:8]
            assert rel(rows, vec[f"out/{g}/{kind}/{s}"]) < 1e-3
            assert abs(float(t.double().norm()) - float(vec[f"outnorm/{g}/{kind}/{s}"][0])) < 1e-3 * float(
                vec[f"outnorm/{g}/{kind}/{s}"][0])
    # ---- loss
    d = (terms.cpu().double() - terms64).abs()
    bound = 1e-3 * torch.clamp(terms64.abs(), min=1e-2)
    assert bool((d <= bound).all()), (d / bound).max()
    assert abs(loss.item() - float(loss64)) <= 1e-3 * max(abs(float(loss64)), 1e-2)
    # ---- gradients: per tensor against the fp64 oracle, allowance = the reference's own fp32<->fp64 spread
    names = [n for n, _ in named]
    assert names == man["param_keys"]
    pg = [(n, p.grad) for n, p in named]
    assert all(g is not None for _, g in pg)
    box_spread = oc["box_grad"]  # the oracle's own fp32<->fp64 distance per tensor on this machine
    fixture = [vec["spread_grad"]] if "spread_grad" in vec else []
    spread_gate(grad_rels(pg, grads64), names, fixture + [box_spread], f"{case} gradients vs fp64 oracle",
                envelope=() if diverse else other_spreads("spread_grad", case), strict_count=diverse)
    gold_norm = dict(zip(man["param_keys"], vec["grad_norm"]))
    rn = np.array([abs(float(g.double().norm()) - gold_norm[n]) / (gold_norm[n] + 1e-30) for n, g in pg])
    allow_n = np.maximum(1e-3, 2 * np.maximum(box_spread, fixture[0] if fixture else 0))
    print(f"[{case}] gradient norms vs golden: median {np.median(rn):.2e} max {rn.max():.2e}; "
          f"{int((rn > allow_n).sum())} beyond the per-tensor allowance")
    assert np.median(rn) < 1e-3
    # ---- BatchNorm running statistics: two updates per step, in view order
    sd_now = model.state_dict()
    for k, v in sd64.items():
        if k.endswith("running_mean") or k.endswith("running_var"):
            assert torch.allclose(sd_now[k].cpu().double(), v, rtol=1e-3, atol=1e-5), k
        if k.endswith("num_batches_tracked"):
            assert int(sd_now[k]) == int(v) == 2
    for key in [k[3:-12] for k in vec if k.startswith("bn/") and k.endswith("/running_var")]:
        assert np.allclose(sd_now[key + ".running_var"].cpu().numpy(), vec[f"bn/{key}/running_var"], rtol=1e-4,
                           atol=1e-6), key
    if not do_adam:
        return
    # ---- optimizer step on the product's gradients (torch Adam, as the reference loop does): updated weights
    opt.step()
    torch.cuda.synchronize()
    box_step = oc["box_step"]
    fixture = [vec["spread_step"]] if "spread_step" in vec else []
    gfix = [vec["spread_grad"]] if "spread_grad" in vec else []
    updated_weights_gate(named, sd0, sd64, grads64, lr, fixture + [box_step], gfix + [box_spread],
                         f"{case} updated weights vs fp64 oracle")


def test_step_parity_r18_b8_s64(hip_lib):
    run_reference_loop_case("r18_b8_s64")


def test_step_parity_r18_b8_s224_config1(hip_lib):
    """BASELINE config 1 at its stated size: ResNet-18 dual-stream, 8 tile pairs of 224x224, fp32"""
    run_reference_loop_case("r18_b8_s224")


def test_step_parity_r18_b16_s64_diverse(hip_lib):
    """well-conditioned inputs (oracle.diverse_batch): the reference's own fp32<->fp64 spread has a median of ~1e-5
    here and 3/4 of its gradient tensors sit below 1e-3, so the per-tensor gate bites at the north-star 1e-3 and rule 2's
    count bound (2 x the reference's count) is below the number of tensors"""
    run_reference_loop_case("r18_b16_s64_div")


def test_step_parity_r50_b8_s64_diverse(hip_lib):
    """the ResNet-50-DERIVED model that configs 2-4 and bench.py run (heads at x4 widths, 1.665 B parameters, fuser GEMMs
    up to 18432 x 18432) on well-conditioned inputs: forward, loss and every gradient against the derived
    oracle.  (The round-2 case on N(0,1) pixels, r50_b8_s64, is retired from the GPU suite: on it the reference's own
    fp32 run sits 2e-2 from its fp64 run and the gate could not fail -- VERDICT r2 weak #2; its fixture remains for
    tests/test_oracle.py.)"""
    run_reference_loop_case("r50_b8_s64_div")


def test_cpu_tensors_fail_loudly(hip_lib):
    from msf_wsi_amd._lib import MsfwsiHipError

    model = build_product("resnet18")
    x = torch.zeros(1, 3, 64, 64)
    with pytest.raises(MsfwsiHipError):
        model.context_encoder(x)


def test_structure_small_batch(hip_lib):
    """B=2 plumbing case (BatchNorm1d over 2 rows is chaotic, so only structure and the loss are checked)"""
    from oracle import msfwsi_oracle as orc

    vec, man = load_golden("r18_b2_s64")
    model = build_product("resnet18").cuda().train()
    (c1, c2), (t1, t2), idx = orc.synthetic_batch(man["B"], man["size"], 16, man["data_seed"])
    with torch.no_grad():
        outs = model((c1.cuda(), t1.cuda()), (c2.cuda(), t2.cuda()), idx)
    assert len(outs) == 3 and all(len(g) == 4 and all(len(t) == 4 for t in g) for g in outs)
    fo = flat_outputs(outs)
    dims = {"context": (2, [64, 128, 256, 512]), "target": (32, [64, 128, 256, 512]),
            "fuser": (2, [576, 1152, 2304, 4608])}
    for (g, kind, s), t in fo.items():
        assert tuple(t.shape) == (dims[g][0], dims[g][1][s]) and not t.requires_grad
    loss, _ = reference_loop_loss(outs)
    assert abs(loss.item() - float(vec["loss"][0])) < 5e-3
    with pytest.raises(AssertionError):  # the reference's only in-path assertion (backbone.py:152)
        model((c1.cuda(), t1.cuda()), (c2.cuda(), t2.cuda()), [idx[0][:, :8], idx[1]])
