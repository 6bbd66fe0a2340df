"""GPU box: per-tensor distance of the product's fp32 ResNet-50 trunk gradients from the fp64 oracle on the
well-conditioned trunk case, in backward order -- where does the error enter (a jump = a gate flip; a ramp = arithmetic)?
    [MSFWSI_FOLD_BN3=0 MSFWSI_FOLD_BN3_FWD=0 ...] python tools/trunk_diag.py"""
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
    vec, man = load_golden("r50enc_b16_s64_div")
    enc, sd0, x, Rs = _trunk_case(man, int(os.environ.get("TRUNK_SEED", man["data_seed"])))
    f64, g64 = _trunk_oracle(sd0, x, Rs)
    enc = enc.cuda().train()
    feats = enc(x.cuda())
    loss = sum((f * r.cuda()).sum() for f, r in zip(feats, Rs))
    loss.backward()
    torch.cuda.synchronize()
    named = dict(enc.named_parameters())
    print("features rel", [f"{rel(f.float(), r):.1e}" for f, r in zip(feats, f64)], "reference fp32:", vec["spread_feat"])
    names = man["param_keys"]
    rels = np.array([rel(named[k].grad.double().cpu(), g64[k]) for k in names])
    print(f"median {np.median(rels):.2e} p90 {np.quantile(rels, .9):.2e} max {rels.max():.2e}; reference median "
          f"{np.median(vec['spread_grad']):.2e}")
    for k, r, s in list(zip(names, rels, vec["spread_grad"]))[::-1]:
        print(f"{r:.2e}  (ref {s:.1e})  {k}")


if __name__ == "__main__":
    main()
