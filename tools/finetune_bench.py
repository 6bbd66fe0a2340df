#!/usr/bin/env python
"""BASELINE config 5 (the fine-tune path) timed on one MI355X: HookNet (two ResNet U-Nets + hook), Dice loss on both
logit maps, backward, torch Adam -- the reference loop's statements (tools/ssl_finetune.py:441-458) at the recipe's batch
(scripts/bcss.sh: -b 64, 256x256 tiles, --amp).  Not the headline metric (bench.py is); prints one JSON line.

    python tools/finetune_bench.py [--arch resnet18 --batch 64 --size 256 --steps 20 --warmup 5 --dtype bf16]"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--arch", default="resnet18")
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--classes", type=int, default=5)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp16", "fp32"])
    args = ap.parse_args()
    from msf_wsi_amd import losses
    from msf_wsi_amd.models.hooknet import HookNet

    torch.manual_seed(3407)
    model = HookNet(encoder_name=args.arch, encoder_weights=None, classes=args.classes + 1).cuda().train()
    g = torch.Generator(device="cuda").manual_seed(0)
    x1 = torch.randn(args.batch, 3, args.size, args.size, generator=g, device="cuda")
    x2 = torch.randn(args.batch, 3, args.size, args.size, generator=g, device="cuda")
    m1 = torch.randint(0, args.classes + 1, (args.batch, args.size, args.size), generator=g, device="cuda")
    m2 = torch.randint(0, args.classes + 1, (args.batch, args.size, args.size), generator=g, device="cuda")
    criterion = losses.DiceLoss(losses.MULTICLASS_MODE, classes=list(range(1, args.classes + 1)), from_logits=True)
    opt = torch.optim.Adam(model.parameters(), 1e-3)
    amp = args.dtype != "fp32"
    scaler = torch.amp.GradScaler("cuda", enabled=amp)
    dt = {"bf16": torch.bfloat16, "fp16": torch.float16, "fp32": torch.float32}[args.dtype]

    def step():
        with torch.autocast("cuda", enabled=amp, dtype=dt if amp else torch.bfloat16):
            c, t = model(x1, x2)
            loss = 0.25 * criterion(c, m1) + 0.75 * criterion(t, m2)
        opt.zero_grad()
        scaler.scale(loss).backward()
        scaler.step(opt)
        scaler.update()
        return loss

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    torch.cuda.synchronize()
    dt_s = time.perf_counter() - t0
    print(json.dumps({"metric": "fine-tune tile pairs/s (config 5: HookNet, Dice, Adam)", "value": round(args.batch * args.steps / dt_s, 2),
                      "ms_per_step": round(1e3 * dt_s / args.steps, 2), "arch": args.arch, "batch": args.batch,
                      "size": args.size, "dtype": args.dtype, "loss": float(loss),
                      "peak_mem_GiB": round(torch.cuda.max_memory_allocated() / 2 ** 30, 2)}))


if __name__ == "__main__":
    main()
