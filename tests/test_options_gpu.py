"""GPU: every dispatch / schedule option of the engine (Engine.OPTIONS -- the attributes that replaced round 5's 57 MSFWSI_*
environment switches) flipped ONE AT A TIME away from its default, and the run held to the SAME parity gate as the default:

  * block- and trunk-level options on the ResNet-50 trunk at 224 x 224 (fixture r50enc_b8_s224_div, bf16: the 16-bit-only
    image / panel kernels dispatch here; production size gates lifted as in test_headline_geometry_gpu): features and every
    gradient tensor within 2 x the distance of the REFERENCE UNDER AUTOCAST from its fp64 run;
  * step-level options (head pairing, stored head gradients, gradient buckets, the streams of the heads and context passes)
    on the fused ResNet-18 step in fp32 (fixture r18_b8_s64): loss 1e-3, updated weights through the Adam-aware gate.
    (`coalesce_views` -- the lockstep pairing of the two views when BatchNorm statistics are exchanged -- acts only with
    more than one rank: its two sides are the 4-rank and 2-rank cases of tests/test_dist_gpu.py; flipped here it must at
    least leave the one-rank step unchanged.)

So no A/B loser that is still reachable is an untested product path.  An option whose flipped
side is invalid on its own is listed with the options it must be flipped with."""
import numpy as np
import pytest
import torch

from helpers import LOWP_FLOOR, LR, build_product, gate_updated_weights, load_golden, lowp_gate, oracle_case, rel
from test_encoder_gpu import _trunk_case, _trunk_oracle, _trunk_product

pytestmark = pytest.mark.gpu

TRUNK_CASE = "r50enc_b8_s224_div"
STEP_CASE = "r18_b8_s64"

# option -> the value it is flipped to (booleans: the opposite of the default) and the options that go with it
TRUNK_FLIPS = {
    "halo3x3": {}, "fuse_pro3x3": {}, "fuse_gate": {}, "gate_bits": {}, "fuse_two_source": {}, "dgrad2_pro": {}, "panel_fwd": {},
    "panel_dgrad": {}, "img3x3": {}, "img3x3_layer1": {}, "img3x3_s2": {}, "gap_stride_fused": {}, "fuse_a2_wgrad": {},
    "panel_gram": {}, "stem_run": {}, "stem_s2d": {}, "stem_fuse_bnbwd": {}, "lores_resid": {},
    "fold_bn3_fwd": {}, "fold_ds_fwd": {}, "fold_ds_strided": {},
    # the backward fold reads what the forward fold kept (Gram matrix, column sums): off together with the forward's
    "fold_bn3": {"fold_bn3_fwd": False}, "fold_ds": {"fold_ds_fwd": False, "fold_ds_strided": False},
    "fuse_a2_wgrad_max_c": {"fuse_a2_wgrad_max_c": 256}, "img3x3_chunk_bytes": {"img3x3_chunk_bytes": 1 << 30},
    "img3x3_min_fill": {"img3x3_min_fill": 1e9}, "panel_fwd_min_k": {"panel_fwd_min_k": 64},
    "dgrad2_pro_max_c": {"dgrad2_pro_max_c": 512},   # the in-launch normalisation of the second source at every width
}
STEP_FLIPS = ("heads_on_streams", "pair_head_wgrad", "pair_head_fwd", "bucket_inter", "store_head_wgrad", "ctx_stream",
              "coalesce_views")


def test_every_option_is_covered():
    from msf_wsi_amd.engine import Engine

    assert set(TRUNK_FLIPS) | set(STEP_FLIPS) == set(Engine.OPTIONS), set(Engine.OPTIONS) ^ (set(TRUNK_FLIPS) | set(STEP_FLIPS))


_ORACLE = {}


def _trunk_reference():
    if "trunk" not in _ORACLE:
        vec, man = load_golden(TRUNK_CASE)
        enc, sd0, x, Rs = _trunk_case(man)
        f64, g64 = _trunk_oracle(sd0, x, Rs, want_loss=float(vec["loss"][0]))
        _ORACLE["trunk"] = (vec, man, sd0, x, Rs, f64, g64)
    return _ORACLE["trunk"]


@pytest.mark.parametrize("option", sorted(TRUNK_FLIPS))
def test_trunk_option_flipped(hip_lib, reproducible_sums, option):
    from msf_wsi_amd.engine import Engine

    vec, man, sd0, x, Rs, f64, g64 = _trunk_reference()
    enc, _, _, _ = _trunk_case(man)
    eng = Engine()
    eng.img3x3_min_fill, eng.img3x3_chunk_bytes = 0.0, 1   # the production dispatch at eight images (test_headline_geometry_gpu)
    flips = dict(TRUNK_FLIPS[option]) or {}
    if option not in flips:
        assert isinstance(getattr(eng, option), bool), option
        flips[option] = not getattr(eng, option)
    for k, v in flips.items():
        assert k in Engine.OPTIONS and getattr(eng, k) != v, (k, v)
        setattr(eng, k, v)
    enc._engine = eng
    feats, grads = _trunk_product(enc, x, Rs, torch.bfloat16)
    names = man["param_keys"]
    rels = np.array([rel(grads[k], g64[k]) for k in names])
    fr = np.array([rel(f.float(), r) for f, r in zip(feats, f64)])
    assert (fr <= np.maximum(LOWP_FLOOR[torch.bfloat16], 2.0 * vec["spread_feat_bf16"])).all(), (option, fr)
    lowp_gate(rels, names, vec["spread_grad_bf16"], LOWP_FLOOR[torch.bfloat16], f"trunk 224 bf16 with {flips}")


@pytest.mark.parametrize("option", STEP_FLIPS)
def test_step_option_flipped(hip_lib, option):
    from msf_wsi_amd.train import PretrainStep

    oc = oracle_case(STEP_CASE)
    (c1, c2), (t1, t2), idx = oc["batch"]
    batch = ((c1.cuda(), c2.cuda()), (t1.cuda(), t2.cuda()), idx)

    def trainer():
        model = build_product("resnet18").cuda().train()
        ts = PretrainStep(model, lr=LR, global_batch=oc["B"], dtype=torch.float32, use_scaler=False)
        assert isinstance(getattr(ts.engine, option), bool)
        setattr(ts.engine, option, not getattr(ts.engine, option))
        return model, ts

    # the multi-stream schedule (three streams, heads on them) starts with the SECOND step of a shape: the first one
    # calibrates the memory plan on one stream.  A throw-away trainer takes that step; the trainer under test inherits its
    # calibration, so that its FIRST step -- the one the fp64 oracle of the fixture describes -- runs the flipped schedule
    _, warm = trainer()
    warm.step(batch)
    torch.cuda.synchronize()
    model, ts = trainer()
    for attr in ("_calib", "_dual_ok", "_dual_started"):
        setattr(ts.engine, attr, getattr(warm.engine, attr))
    del warm
    loss = float(ts.step(batch))
    assert ",dual-stream" in ts.engine.last_plan, ts.engine.last_plan
    torch.cuda.synchronize()
    assert abs(loss - oc["loss64"]) <= 1e-3 * max(abs(oc["loss64"]), 1e-2), (option, loss, oc["loss64"])
    gate_updated_weights(list(model.named_parameters()), STEP_CASE, f"fused step with {option} flipped")
