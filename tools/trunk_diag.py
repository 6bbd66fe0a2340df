"""GPU box: per-tensor distance of the product's fp32 ResNet-50 trunk gradients from the fp64 oracle on the
well-conditioned trunk case, in backward order -- where does the error enter (a jump = a gate flip; a ramp = arithmetic)?
    [MSFWSI_ENGINE=fold_bn3=0,fold_bn3_fwd=0,...] python tools/trunk_diag.py"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import load_golden, rel  # noqa: E402
from test_encoder_gpu import _trunk_case, _trunk_oracle  # noqa: E402


def main():
    vec, man = load_golden(os.environ.get("TRUNK_CASE", "r50enc_b16_s64_div"))
    enc, sd0, x, Rs = _trunk_case(man, int(os.environ.get("TRUNK_SEED", man["data_seed"])))
    f64, g64 = _trunk_oracle(sd0, x, Rs)
    f32, g32 = _trunk_oracle({k: v.clone() for k, v in sd0.items()}, x, Rs, torch.float32)  # the oracle's own fp32 run, this host
    enc = enc.cuda().train()
    feats = enc(x.cuda())
    loss = sum((f * r.cuda()).sum() for f, r in zip(feats, Rs))
    loss.backward()
    torch.cuda.synchronize()
    named = dict(enc.named_parameters())
    print("features rel", [f"{rel(f.float(), r):.1e}" for f, r in zip(feats, f64)], "oracle fp32 here:",
          [f"{rel(a, b):.1e}" for a, b in zip(f32, f64)], "reference fp32 (fixture):", vec["spread_feat"])
    names = man["param_keys"]
    rels = np.array([rel(named[k].grad.double().cpu(), g64[k]) for k in names])
    box = np.array([rel(g32[k], g64[k]) for k in names])
    print(f"median {np.median(rels):.2e} p90 {np.quantile(rels, .9):.2e} max {rels.max():.2e}; oracle fp32 here: median "
          f"{np.median(box):.2e} max {box.max():.2e}; reference (fixture) median {np.median(vec['spread_grad']):.2e}")
    if os.environ.get("TRUNK_BRIEF", "0") != "0":
        return
    for k, r, s, b in list(zip(names, rels, vec["spread_grad"], box))[::-1]:
        print(f"{r:.2e}  (ref {s:.1e}, oracle fp32 here {b:.1e})  {k}")


if __name__ == "__main__":
    main()
