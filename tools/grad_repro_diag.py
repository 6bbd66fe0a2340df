"""GPU box: which gradient tensors of the ResNet-50 trunk are NOT bit-for-bit reproducible between two identical runs (same
weights, same inputs) under msfwsi_set_tuning(15, 1)?  Printed in backward order (from the loss down): the first tensor that
differs names the launch whose accumulation order is not fixed.   python tools/grad_repro_diag.py [case] [dtype]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import load_golden  # noqa: E402
from test_encoder_gpu import _trunk_case  # noqa: E402
from msf_wsi_amd import _lib  # noqa: E402


def main():
    case = sys.argv[1] if len(sys.argv) > 1 else "r50enc_b16_s64_div"
    dt = {"bf16": torch.bfloat16, "fp16": torch.float16, "fp32": torch.float32}[sys.argv[2] if len(sys.argv) > 2 else "bf16"]
    _lib.load().msfwsi_set_tuning(15, 1)
    vec, man = load_golden(case)
    runs = []
    for r in range(3):
        enc, sd0, x, Rs = _trunk_case(man)
        enc = enc.cuda().train()
        with torch.autocast("cuda", dtype=dt, enabled=dt != torch.float32):
            feats = enc(x.cuda())
        loss = sum((f.float() * rr.cuda()).sum() for f, rr in zip(feats, Rs))
        loss.backward()
        torch.cuda.synchronize()
        runs.append(({k: p.grad.detach().clone() for k, p in enc.named_parameters() if p.grad is not None},
                     [f.detach().clone() for f in feats]))
    names = list(runs[0][0])[::-1]
    print("features bitwise equal:", all(torch.equal(a, b) for a, b in zip(runs[0][1], runs[1][1])))
    nbad = 0
    for k in names:
        d = [float((runs[i][0][k].double() - runs[0][0][k].double()).norm() / (runs[0][0][k].double().norm() + 1e-300)) for i in (1, 2)]
        if max(d) > 0:
            nbad += 1
            if nbad <= 12:
                print(f"  {k}: {d[0]:.2e} {d[1]:.2e}")
    print(f"{nbad}/{len(names)} gradient tensors differ between runs")


if __name__ == "__main__":
    main()
