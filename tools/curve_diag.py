"""GPU box: the product's 30-step loss curves (fp32 / bf16 / fp16 fused step) beside the reference's curves of
tests/golden/r18_b16_s64_curve (fp32, fp64, bf16 / fp16 autocast) -- how far apart do trajectories of this chaotic
system drift, per precision?      python tools/curve_diag.py [dtype ...]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import LR, build_case, load_golden  # noqa: E402
from msf_wsi_amd.train import PretrainStep  # noqa: E402
from oracle import msfwsi_oracle as orc  # noqa: E402


def main():
    vec, man = load_golden("r18_b16_s64_curve")
    B, size, steps = man["B"], man["size"], man["steps"]
    ref = vec["loss_fp32"]
    print("ref fp32 ", " ".join(f"{v:.3f}" for v in ref))
    for tag in ("fp64", "bf16"):
        print(f"ref {tag} d", " ".join(f"{v:.3f}" for v in vec["loss_" + tag] - ref))
    print("oracle32 d", " ".join(f"{v:.3f}" for v in vec["oracle_fp32_dev"]))
    for name in (sys.argv[1:] or ["fp32", "bf16", "bf16", "fp16"]):
        if name.startswith("oracle"):  # the oracle under autocast on THIS machine's CPU: another reference-autocast sample
            torch.set_num_threads(max(1, (os.cpu_count() or 2) // 2))
            sd = {k: v.detach().clone() for k, v in build_case(man).state_dict().items()}
            lr = orc.init_lr(LR, B)
            opt = orc.Adam(sd, [lr, lr, lr])
            ac = {"oracle-bf16": torch.bfloat16, "oracle-fp32": None}[name]
            ls = []
            for t in range(steps):
                l_, _, _, _ = orc.train_step(sd, orc.diverse_batch(B, size, 16, man["curve_seed0"] + t), opt, autocast_dtype=ac)
                ls.append(float(l_))
            ls = np.array(ls)
            print(f"{name} d", " ".join(f"{v:.3f}" for v in ls - ref), f"| last-10 mean {ls[-10:].mean():.4f} vs {ref[-10:].mean():.4f}")
            continue
        scaler = None
        if name.endswith("-noscale"):
            name, scaler = name[:-8], False
        dt = {"fp32": torch.float32, "bf16": torch.bfloat16, "fp16": torch.float16}[name]
        model = build_case(man).cuda().train()
        ts = PretrainStep(model, lr=LR, global_batch=B, dtype=dt, init_scale=1024.0 if dt == torch.float16 else 65536.0,
                          use_scaler=scaler)
        ls = []
        for t in range(steps):
            (c1, c2), (t1, t2), idx = orc.diverse_batch(B, size, 16, man["curve_seed0"] + t)
            ls.append(ts.step(((c1.cuda(), c2.cuda()), (t1.cuda(), t2.cuda()), idx)))
        ls = torch.stack(ls).cpu().numpy().ravel()
        print(f"prod {name} d", " ".join(f"{v:.3f}" for v in ls - ref), f"| last-10 mean {ls[-10:].mean():.4f} vs {ref[-10:].mean():.4f}"
              f" | Adam steps applied {ts.t}/{steps}, loss scale {ts.scale.item():.0f}")


if __name__ == "__main__":
    main()
