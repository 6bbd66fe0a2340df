"""CPU: the oracle (oracle/msfwsi_oracle.py) against the golden vectors generated from the REAL reference
(tests/golden/make_golden.py).  This is what pins the oracle; the GPU parity tests then compare the product
with the oracle.  Runs without a GPU and without /root/reference."""
import numpy as np
import pytest
import torch

from helpers import LR, WEIGHTS, build_case, build_product, case_batch, load_golden, rel


@pytest.fixture(scope="module", params=["r18_b8_s64", "r18_b16_s64_div"])
def case(request):
    from oracle import msfwsi_oracle as orc

    vec, man = load_golden(request.param)
    model = build_case(man)
    sd0 = {k: v.detach().clone() for k, v in model.state_dict().items()}
    B = man["B"]
    batch = case_batch(man)
    osd = {k: (v.double() if v.is_floating_point() else v.clone()) for k, v in sd0.items()}
    (c1, c2), (t1, t2), idx = batch
    b64 = ((c1.double(), c2.double()), (t1.double(), t2.double()), idx)
    lr = orc.init_lr(LR, B)
    loss, terms, outs, grads = orc.train_step(osd, b64, orc.Adam(osd, [lr, lr, lr]), 4, 0.5, WEIGHTS)
    return vec, man, sd0, osd, loss, terms, outs, grads


def test_seeded_init_matches_reference(case):
    vec, man, sd0 = case[0], case[1], case[2]
    assert [k for k in sd0] == [k for k, _, _ in man["keys"]]
    for (k, shape, dt), v in zip(man["keys"], sd0.values()):
        assert list(v.shape) == shape and str(v.dtype) == dt, k
    assert np.allclose([float(v.double().sum()) for v in sd0.values()], vec["init_sum"], rtol=1e-12, atol=1e-12)
    assert np.allclose([float(v.double().abs().sum()) for v in sd0.values()], vec["init_abs"], rtol=1e-12, atol=1e-12)


def test_oracle_loss_and_outputs_match_reference(case):
    vec, man, _, _, loss, terms, outs, _ = case
    flat = torch.stack([t for row in terms for t in row])
    assert torch.allclose(flat, torch.as_tensor(vec["terms"]), rtol=0, atol=1e-9)
    assert abs(float(loss) - float(vec["loss"][0])) < 1e-9
    names = ["p1", "p2", "z1", "z2"]
    for gi, g in enumerate(("context", "target", "fuser")):
        for ti in range(4):
            for s in range(4):
                t = outs[gi][ti][s]
                rows = t if g != "target" else t[:: max(1, t.shape[0] // 8)][:8]
                assert rel(rows, vec[f"out/{g}/{names[ti]}/{s}"]) < 1e-6
                assert t.requires_grad == (ti < 2)


def test_oracle_gradients_match_reference(case):
    vec, man, _, _, _, _, _, grads = case
    norms = np.array([float(grads[k].double().norm()) for k in man["param_keys"]])
    sums = np.array([float(grads[k].double().sum()) for k in man["param_keys"]])
    assert np.allclose(norms, vec["grad_norm"], rtol=1e-7, atol=1e-12)
    assert np.allclose(sums, vec["grad_sum"], rtol=1e-6, atol=1e-9)
    for k in ("context_encoder.bn1.weight", "context_encoder.bn1.bias"):
        assert rel(grads[k], vec["grad/" + k]) < 1e-5  # stored as fp32


def test_oracle_adam_and_running_stats_match_reference(case):
    vec, man, sd0, osd = case[0], case[1], case[2], case[3]
    step = np.array([float((osd[k].double() - sd0[k].double()).norm()) for k in man["param_keys"]])
    assert np.allclose(step, vec["step_norm"], rtol=1e-6, atol=1e-12)
    assert np.allclose([float(osd[k].double().sum()) for k in man["param_keys"]], vec["w1_sum"], rtol=1e-9, atol=1e-9)
    for key in ("context_encoder.bn1", "target_encoder.layer2.0.downsample.1", "target_encoder.layer4.1.bn2",
                "inter_projector.0.1", "context_predictor.3.1"):
        for stat in ("running_mean", "running_var"):  # means of already-normalised inputs are ~1e-17: abs tol
            assert np.allclose(osd[f"{key}.{stat}"].numpy(), vec[f"bn/{key}/{stat}"], rtol=1e-6, atol=1e-7)
        assert int(osd[key + ".num_batches_tracked"]) == int(vec[f"bn/{key}/nbt"][0]) == 2


def test_oracle_fp32_golden_c1_config():
    """config 1 of BASELINE.json (ResNet-18, B=8, 224x224, fp32): loss of the real reference, fp32"""
    vec, man = load_golden("r18_b8_s224")
    assert man["B"] == 8 and man["size"] == 224 and abs(float(vec["loss"][0]) - 0.036828268) < 1e-6


def test_param_groups_and_lr():
    from oracle import msfwsi_oracle as orc

    model = build_product("resnet18")
    sd = model.state_dict()
    groups = orc.param_groups(sd)
    assert [len(g) for g in groups] == [108, 108, 48]
    assert sum(sd[k].numel() for g in groups for k in g) == 123551584
    assert abs(orc.init_lr(1e-3, 8) - 1e-3 * 8 ** 0.5 / 32 ** 0.5) < 1e-15


@pytest.mark.parametrize("case_name", ["r50_b8_s64", "r50_b2_s64"])
def test_resnet50_derived_fixture_manifest(case_name):
    """the ResNet-50-DERIVED model (reference trunk + reference head factories + reference forward, width
    list x block expansion) that BASELINE configs 2-4 run: the fixture written by make_golden.py after it asserted
    oracle == reference and product init == reference init (the 1.665 B-parameter model is not rebuilt here)"""
    vec, man = load_golden(case_name)
    assert man["arch"] == "resnet50" and "derived oracle" in man["provenance"]
    assert len(man["keys"]) == 924 and len(man["param_keys"]) == 462
    n_param = sum(int(np.prod(shape)) for k, shape, _ in man["keys"] if k in set(man["param_keys"]))
    assert abs(n_param - 1.665e9) < 1e6
    assert vec["terms"].shape == (12,) and np.isfinite(vec["terms"]).all()
    w = np.tile(np.array(WEIGHTS), 3)
    assert abs(float((vec["terms"] * w).sum()) - float(vec["loss"][0])) < 1e-7  # fp32 case: terms summed in fp32
    assert vec["grad_norm"].shape == (462,) and (vec["grad_norm"] >= 0).all()
    if "spread_grad" in vec:  # the reference's own fp32<->fp64 noise floor on this model: ~2e-2 on most tensors
        assert vec["spread_grad"].shape == (462,) and np.median(vec["spread_grad"]) > 1e-3


def test_updated_weights_gate_accepts_reference_noise_and_rejects_a_wrong_step():
    """the gate of the GPU parity tests, exercised on the CPU: the reference arithmetic in fp32 against fp64 passes
    (its Adam sign flips are the Adam steps of gradients within tolerance); an update that is wrong on a tenth of one
    tensor, or a step with the wrong learning rate, does not"""
    from helpers import oracle_case, updated_weights_gate

    oc = oracle_case("r18_b8_s64")
    sp, gsp = [oc["vec"]["spread_step"]], [oc["vec"]["spread_grad"]]
    named = [(n, oc["sd32"][n]) for n in oc["names"]]
    updated_weights_gate(named, oc["sd0"], oc["sd64"], oc["grads64"], oc["lr"], sp, gsp, "fp32 oracle vs fp64")
    k = oc["names"].index("context_encoder.layer3.0.conv1.weight")
    bad = list(named)
    t = named[k][1].clone().contiguous()
    t.view(-1)[: t.numel() // 10] += 2 * oc["lr"]
    bad[k] = (named[k][0], t)
    with pytest.raises(AssertionError):
        updated_weights_gate(bad, oc["sd0"], oc["sd64"], oc["grads64"], oc["lr"], sp, gsp, "corrupted tensor")
    half = [(n, oc["sd0"][n].double() + 0.5 * (oc["sd64"][n].double() - oc["sd0"][n].double())) for n in oc["names"]]
    with pytest.raises(AssertionError):
        updated_weights_gate(half, oc["sd0"], oc["sd64"], oc["grads64"], oc["lr"], sp, gsp, "half learning rate")


def test_diverse_fixture_is_well_conditioned():
    """the well-conditioned case (oracle.diverse_batch) does what it is for: the REFERENCE's own fp32 run sits at a
    median of ~1e-5 from its fp64 run (N(0,1) pixels: 1e-3 .. 2e-2), most gradient tensors are below the north-star
    1e-3, and rule 2 of spread_gate (count <= 2 x the reference's count) is below the tensor count, i.e. can trip"""
    vec, man = load_golden("r18_b16_s64_div")
    sp = vec["spread_grad"]
    assert man["input_kind"] == "diverse" and np.median(sp) < 1e-4 and (sp <= 1e-3).mean() >= 0.7
    assert 2 * int((sp > 1e-3).sum()) < len(sp)
    for tag in ("bf16", "fp16"):  # the reference under autocast: per-tensor yardsticks of the 16-bit runs
        assert vec[f"spread_grad_{tag}"].shape == sp.shape and vec[f"spread_out_{tag}"].shape == (48,)
        assert vec[f"spread_terms_{tag}"].shape == (12,)
    assert vec["spread_out_fp16"].max() < vec["spread_out_bf16"].max() < 0.2


def test_oracle_under_autocast_is_a_sample_of_the_reference_under_autocast():
    """the oracle's autocast mode against the fixture written from the REAL reference under
    torch.autocast("cpu", bfloat16).  In one process the two have the same forward bit for bit (asserted by
    make_golden.py); across processes / machines oneDNN's bf16 kernels block their sums differently, so here the
    oracle's bf16 loss terms are held to the fixture's bf16 distance from fp64 (x3: a sample against a sample)"""
    from oracle import msfwsi_oracle as orc

    vec, man = load_golden("r18_b16_s64_div")
    model = build_case(man)
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    nop = type("NoOpt", (), {"step": lambda self, *a, **k: None})()
    loss, terms, _, grads = orc.train_step(sd, case_batch(man), nop, 4, 0.5, WEIGHTS, autocast_dtype=torch.bfloat16)
    t16 = np.array([float(t) for row in terms for t in row])
    d = np.abs(t16 - vec["terms"])
    assert (d <= np.maximum(1e-4, 3.0 * vec["spread_terms_bf16"])).sum() >= 11 and d.max() <= 3.0 * vec["spread_terms_bf16"].max()
    assert all(g is not None and g.dtype == torch.float32 for g in grads.values())  # master gradients stay fp32


def test_lowp_gate_passes_the_reference_sample_and_rejects_a_scaled_error():
    from helpers import lowp_gate

    vec, man = load_golden("r18_b16_s64_div")
    ref = vec["spread_grad_bf16"]
    names = man["param_keys"]
    lowp_gate(ref, names, ref, 2.0 ** -7, "the reference's own sample")
    with pytest.raises(AssertionError):
        lowp_gate(np.minimum(3.0 * ref, 2.0), names, ref, 2.0 ** -7, "three times the reference's noise")
    worse = ref.copy()
    worse[: len(ref) // 8] = 2.5 * ref.max()
    with pytest.raises(AssertionError):
        lowp_gate(worse, names, ref, 2.0 ** -7, "an eighth of the tensors far out")
