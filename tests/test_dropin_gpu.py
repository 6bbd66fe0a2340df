"""GPU: the reference's own wrapping and loop statements around the drop-in module
(tools/ssl_train.py:160-170 SyncBatchNorm + DDP, :441-474 autocast + GradScaler + backward + step)."""
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn as nn

from helpers import LR, build_product, gate_updated_weights, oracle_case, reference_loop_loss

pytestmark = pytest.mark.gpu
CASE = "r18_b8_s64"  # golden case (its fixture carries the reference's own fp32<->fp64 spread per tensor)
B, SIZE, K = 8, 64, 16


def _batch(seed=0):
    from oracle import msfwsi_oracle as orc

    return orc.synthetic_batch(B, SIZE, K, seed)


@pytest.mark.parametrize("amp_dtype", [torch.bfloat16, torch.float16])
def test_reference_amp_loop(hip_lib, amp_dtype):
    """`--amp [--bf16]` path: autocast(fp16 | bf16) + GradScaler around the module, three iterations"""
    model = build_product("resnet18").cuda().train()
    named = list(model.named_parameters())
    groups = [[p for n, p in named if n.startswith(pre)] for pre in ("context_", "target_", "inter_")]
    opt = torch.optim.Adam([{"params": g, "lr": 5e-4} for g in groups], lr=5e-4)
    scaler = torch.amp.GradScaler("cuda", enabled=True)
    (c1, c2), (t1, t2), idx = _batch()
    c1, c2, t1, t2 = c1.cuda(), c2.cuda(), t1.cuda(), t2.cuda()
    losses = []
    for _ in range(3):
        with torch.autocast("cuda", enabled=True, dtype=amp_dtype):
            outputs = model((c1, t1), (c2, t2), idx)
            loss, _ = reference_loop_loss(outputs)
        assert outputs[0][0][0].dtype == amp_dtype
        opt.zero_grad()
        scaler.scale(loss).backward()
        scaler.step(opt)
        scaler.update()
        losses.append(loss.item())
    assert all(torch.isfinite(torch.tensor(losses)))
    if amp_dtype == torch.bfloat16:
        assert scaler.get_scale() == 65536.0
        assert all(p.grad is not None and torch.isfinite(p.grad).all() for _, p in named)
        assert losses[2] < losses[0]  # three Adam steps on a fixed batch reduce the loss
    else:  # fp16 may overflow at the initial scale: GradScaler then skips the step and backs off
        assert scaler.get_scale() in (65536.0, 32768.0, 16384.0, 8192.0)
        assert losses[2] <= losses[0]


def _ddp_worker(rank, world, port, ret):
    from msf_wsi_amd.dist import shard_range

    torch.cuda.set_device(0)
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    try:
        model = build_product("resnet18")
        model = nn.SyncBatchNorm.convert_sync_batchnorm(model)  # ssl_train.py:160
        model.cuda(0)
        ddp = nn.parallel.DistributedDataParallel(model, device_ids=[0])  # ssl_train.py:170
        named = list(ddp.module.named_parameters())
        groups = [[p for n, p in named if n.startswith(pre)] for pre in ("context_", "target_", "inter_")]
        lr = LR * (B ** 0.5) / (32 ** 0.5)
        opt = torch.optim.Adam([{"params": g, "lr": lr} for g in groups], lr=lr)
        (c1, c2), (t1, t2), idx = _batch()
        lo, hi = shard_range(B, world, rank)
        ddp.train()
        outputs = ddp((c1[lo:hi].cuda(), t1[lo * K:hi * K].cuda()), (c2[lo:hi].cuda(), t2[lo * K:hi * K].cuda()),
                      [idx[0][lo:hi], idx[1][lo:hi]])
        loss, _ = reference_loop_loss(outputs)
        opt.zero_grad()
        loss.backward()
        opt.step()
        torch.cuda.synchronize()
        lsum = torch.tensor([loss.item()], dtype=torch.float64)
        dist.all_reduce(lsum)
        if rank == 0:
            ret["loss"] = float(lsum) / world
            ret["sd"] = {k: v.detach().cpu() for k, v in ddp.state_dict().items()}
    finally:
        dist.destroy_process_group()


def test_syncbn_ddp_wrapping_matches_single_process(hip_lib):
    """the reference's own wrapping -- convert_sync_batchnorm + DistributedDataParallel + torch Adam (ssl_train.py:160,
    170, 309) -- around the drop-in module on two ranks == the fp64 oracle on the full batch"""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ret = mp.get_context("spawn").Manager().dict()
    mp.spawn(_ddp_worker, args=(2, port, ret), nprocs=2, join=True)

    oc = oracle_case(CASE)
    assert abs(ret["loss"] - oc["loss64"]) <= 1e-3 * max(abs(oc["loss64"]), 1e-2)
    sd2 = ret["sd"]
    assert all(k.startswith("module.") for k in sd2)  # DDP prefix, as saved by the reference (ssl_train.py:380)
    for k, v in oc["sd64"].items():
        a = sd2["module." + k].double()
        if k.endswith("num_batches_tracked"):
            assert int(a) == int(v) == 2
        elif "running_" in k:
            assert torch.allclose(a, v, rtol=1e-3, atol=1e-5), k
    gate_updated_weights([(n, sd2["module." + n]) for n in oc["names"]], CASE,
                         "SyncBatchNorm + DDP wrapping, 2 ranks: updated weights")
