#!/usr/bin/env python
"""The reference's epoch loop around the fused step -- tools/ssl_train.py:338-392 (epochs, sampler epoch, `save_freq`,
`--resume`) and :408-486 (`train()`), restated minimally on synthetic tiles; no CLI framework, no logger, no dataset I/O.

    python examples/pretrain_loop.py --arch resnet18 --batch 8 --size 64 --epochs 2 --steps-per-epoch 3 --log-dir /tmp/run
    python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 examples/pretrain_loop.py ...   (one rank per GPU)

What differs from the reference's loop, and why: the model is NOT wrapped in SyncBatchNorm / DistributedDataParallel
(`PretrainStep` exchanges the statistics and gradients itself and broadcasts rank 0's weights in its constructor, :160-170),
the loss is read once per epoch (`epoch_loss`, :483-486; the reference's per-step `.item()`, :467, is a host sync), and
`save_checkpoint` is called by EVERY rank, outside the rank-0 guard of :363 (it is a collective; rank 0 writes).
"""
import argparse
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("-a", "--arch", default="resnet18")
    ap.add_argument("-b", "--batch", type=int, default=32, help="GLOBAL batch of tile pairs (ssl_train.py:165 splits it)")
    ap.add_argument("--size", type=int, default=224)
    ap.add_argument("--epochs", type=int, default=2)
    ap.add_argument("--steps-per-epoch", type=int, default=4)
    ap.add_argument("--lr", type=float, default=1e-3)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp16", "fp32"])
    ap.add_argument("--save-freq", type=int, default=1)
    ap.add_argument("--resume", default="")
    ap.add_argument("--log-dir", default="/tmp/msfwsi_run")
    ap.add_argument("--seed", type=int, default=3407)
    args = ap.parse_args(argv)

    from msf_wsi_amd.dist import shard_range
    from msf_wsi_amd.models import resnet
    from msf_wsi_amd.models.backbone import MSFWSI
    from msf_wsi_amd.train import PretrainStep, synthetic_batch

    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    gpu = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(gpu)
    if world > 1:  # ssl_train.py:135-141
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", device_id=torch.device("cuda", gpu))
    if not os.environ.get("MSFWSI_PRETRAINED_DIR"):  # no network on a GPU box: un-pretrained weights stand in (resnet.py:271-274)
        torch.hub.load_state_dict_from_url = lambda url, progress=True, **kw: resnet.__dict__[args.arch]().state_dict()
    torch.manual_seed(args.seed + rank)  # replicas need not start equal: the trainer's constructor broadcasts rank 0's
    model = MSFWSI(resnet.__dict__[args.arch], 4).cuda(gpu).train()                     # ssl_train.py:145-152,163
    dtype = {"bf16": torch.bfloat16, "fp16": torch.float16, "fp32": torch.float32}[args.dtype]
    step = PretrainStep(model, lr=args.lr, global_batch=args.batch, dtype=dtype, arch=args.arch)  # :155,281-310
    start_epoch = 0
    if args.resume:                                                                                # :313-335
        start_epoch = step.resume(torch.load(args.resume, map_location=f"cuda:{gpu}", weights_only=False))
    os.makedirs(args.log_dir, exist_ok=True)
    lo, hi = shard_range(args.batch, world, rank)                                                  # DistributedSampler, :262-275
    for epoch in range(start_epoch, args.epochs):                                                  # :338
        for it in range(args.steps_per_epoch):                                                     # train(), :425
            # sampler.set_epoch(epoch) (:342): the epoch and the step pick the synthetic samples; a rank takes its shard
            (c1, c2), (t1, t2), idx = synthetic_batch(args.batch, args.size, 16, seed=epoch * 100003 + it, device=f"cuda:{gpu}")
            step.step(((c1[lo:hi], c2[lo:hi]), (t1[lo * 16:hi * 16], t2[lo * 16:hi * 16]), [i[lo:hi] for i in idx]))
        loss = step.epoch_loss()                                                                   # :483-486 (collective)
        if rank == 0:
            print(f"epoch {epoch}: loss {loss:.6f}", flush=True)                                  # :363-368
        if (epoch + 1) % args.save_freq == 0 or epoch + 1 == args.epochs:                          # :375-386
            step.save_checkpoint(os.path.join(args.log_dir, f"checkpoint_{epoch:04d}.pth.tar"), epoch)  # every rank
    if world > 1:
        dist.destroy_process_group()
    return step


if __name__ == "__main__":
    main()
