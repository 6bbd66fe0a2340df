"""Diagnostic (GPU box): where does the product's fp32 forward leave the fp64 oracle?  ResNet-18 trunk on the golden
case's context images; per block output, rel-L2 of (product fp32, oracle fp32) against oracle fp64.
    python tools/diag_layers.py [case]"""
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import build_product, load_golden  # noqa: E402
from oracle import msfwsi_oracle as orc  # noqa: E402


def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-300))


def oracle_blocks(sd, prefix, x):
    """encoder_forward of the oracle, returning every block output (NCHW)"""
    outs = []
    y = F.conv2d(x, sd[prefix + "conv1.weight"], None, stride=2, padding=3)
    outs.append(("stem conv", y))
    y = F.relu(orc._bn(sd, prefix + "bn1", y))
    y = F.max_pool2d(y, kernel_size=3, stride=2, padding=1)
    outs.append(("pool", y))
    for s, blocks in enumerate(orc.encoder_layout(sd, prefix), start=1):
        for b, (nconv, has_ds) in enumerate(blocks):
            p = f"{prefix}layer{s}.{b}."
            stride = 2 if (s > 1 and b == 0) else 1
            identity = y
            out = F.conv2d(y, sd[p + "conv1.weight"], None, stride=stride, padding=1)
            outs.append((f"layer{s}.{b}.conv1 raw", out))
            out = F.relu(orc._bn(sd, p + "bn1", out))
            out = F.conv2d(out, sd[p + "conv2.weight"], None, stride=1, padding=1)
            outs.append((f"layer{s}.{b}.conv2 raw", out))
            out = orc._bn(sd, p + "bn2", out)
            if has_ds:
                identity = orc._bn(sd, p + "downsample.1", F.conv2d(y, sd[p + "downsample.0.weight"], None, stride=stride))
            y = F.relu(out + identity)
            outs.append((f"layer{s}.{b} out", y))
        outs.append((f"gap{s}", torch.flatten(F.adaptive_avg_pool2d(y, (1, 1)), 1)))
    return outs


def main():
    case = sys.argv[1] if len(sys.argv) > 1 else "r18_b8_s224"
    vec, man = load_golden(case)
    model = build_product(man["arch"])
    sd0 = {k: v.detach().clone() for k, v in model.state_dict().items()}
    (c1, c2), (t1, t2), idx = orc.synthetic_batch(man["B"], man["size"], 16, man["data_seed"])
    x = c1
    pre = "context_encoder."
    ref = {}
    for tag, dt in (("64", torch.float64), ("32", torch.float32)):
        sd = {k: (v.clone().to(dt) if v.is_floating_point() else v.clone()) for k, v in sd0.items()}
        with torch.no_grad():
            ref[tag] = oracle_blocks(sd, pre, x.to(dt))
    from msf_wsi_amd.engine import Engine

    enc = model.context_encoder.cuda().train()
    eng = Engine()
    eng.update_running = False
    ps = eng.encoder_forward(enc, x.cuda(), torch.float32, save=True)
    torch.cuda.synchronize()
    mine = {"stem conv": ps.stem.c.permute(0, 3, 1, 2), "pool": ps.pooled.permute(0, 3, 1, 2)}
    names = [n for n, _ in enc.named_modules()]
    bi = 0
    for s, stage in enumerate(enc.stages(), start=1):
        for b, _ in enumerate(stage):
            rec = ps.blocks[bi]
            bi += 1
            mine[f"layer{s}.{b}.conv1 raw"] = rec.units[0].c.permute(0, 3, 1, 2)
            mine[f"layer{s}.{b}.conv2 raw"] = rec.units[1].c.permute(0, 3, 1, 2)
            mine[f"layer{s}.{b} out"] = rec.y_out.permute(0, 3, 1, 2)
        mine[f"gap{s}"] = ps.feats[s - 1]
    print(f"{'tensor':28s} {'product fp32':>14s} {'oracle fp32':>14s}   (rel-L2 vs oracle fp64)")
    for (n, a64), (_, a32) in zip(ref["64"], ref["32"]):
        print(f"{n:28s} {rel(mine[n], a64):14.2e} {rel(a32, a64):14.2e}")
    # centred features: what BatchNorm1d of the heads sees
    for s in range(1, 5):
        a64 = dict(ref["64"])[f"gap{s}"]
        c64 = a64 - a64.mean(0, keepdim=True)
        cm = mine[f"gap{s}"].double().cpu()
        cm = cm - cm.mean(0, keepdim=True)
        c32 = dict(ref["32"])[f"gap{s}"].double()
        c32 = c32 - c32.mean(0, keepdim=True)
        print(f"gap{s} centred over the batch: product {rel(cm, c64):.2e}  oracle fp32 {rel(c32, c64):.2e}   "
              f"(|mean|/|centred| = {float(a64.mean(0).norm() / c64.norm() * (a64.shape[0] ** 0.5)):.1f})")


if __name__ == "__main__":
    main()
