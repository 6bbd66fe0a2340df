"""GPU: the product's data-parallel path end to end with two ranks.  The GPU box has ONE card, and RCCL refuses
two ranks on one device, so both ranks share cuda:0 over the gloo backend (it accepts device tensors); the
engine's SyncBN exchange and the per-group gradient reducer run exactly as under RCCL, only the transport
differs.  Claim: 2 ranks x B/2 tile pairs == 1 rank x B tile pairs (loss, updated weights, running stats)."""
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from helpers import LR, build_product, gate_updated_weights, oracle_case

pytestmark = pytest.mark.gpu
CASE = "r18_b8_s64"  # the golden case: its fixture carries the reference's own fp32<->fp64 spread per tensor
B, SIZE, K = 8, 64, 16


def _batch():
    from oracle import msfwsi_oracle as orc

    return orc.synthetic_batch(B, SIZE, K, 0)


def _worker(rank, world, port, ret):
    import os

    from msf_wsi_amd.dist import shard_range
    from msf_wsi_amd.train import PretrainStep

    if world == 2:
        os.environ["MSFWSI_DUAL_STREAM"] = "1"  # the multi-stream schedule under a real multi-rank exchange, from step one
    else:
        # world 4: the other multi-rank schedule -- the two views of an encoder in LOCKSTEP on one stream, one SyncBatchNorm
        # message per BatchNorm and direction for both views (what a rank falls back to when two sets of backward transients
        # do not fit beside the RCCL reserve; MSFWSI_MULTIRANK_STREAMS=0 selects it outright)
        os.environ["MSFWSI_MULTIRANK_STREAMS"] = "0"
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    try:
        model = build_product("resnet18")
        if rank != 0:
            # the reference seeds the parent process only (ssl_train.py:46-48): its spawned workers do NOT build equal
            # replicas, DDP's constructor broadcast (:170) makes them equal.  Every rank but 0 starts from other weights AND
            # other BatchNorm buffers here; the parity with the oracle below (which knows rank 0's weights only) and the
            # num_batches_tracked == 2 check hold only if PretrainStep's rank-0 broadcast ran
            with torch.no_grad():
                g = torch.Generator().manual_seed(500 + rank)
                for p in model.parameters():
                    p.add_(0.05 * torch.randn(p.shape, generator=g))
                for b in model.buffers():
                    b.add_(3)
        model = model.cuda().train()
        # world 2: the all-reduce exchange; world 4: the sharded optimizer (reduce-scatter, Adam on 1/4, all-gather)
        ts = PretrainStep(model, lr=LR, global_batch=B, dtype=torch.float32, use_scaler=False, sync_bn=True,
                          shard_optimizer=(world == 4))
        (c1, c2), (t1, t2), idx = _batch()
        lo, hi = shard_range(B, world, rank)
        local = ((c1[lo:hi].cuda(), c2[lo:hi].cuda()), (t1[lo * K:hi * K].cuda(), t2[lo * K:hi * K].cuda()),
                 [idx[0][lo:hi], idx[1][lo:hi]])
        loss = ts.step(local)
        torch.cuda.synchronize()
        mean_loss = ts.epoch_loss()
        ret[f"collectives{rank}"] = ts.engine.collectives_last_step
        ret[f"plan{rank}"] = ts.engine.last_plan
        ret[f"grad_msgs{rank}"] = ts.reducer.launches_last_step
        ts.sync_master_weights()  # collective (a no-op in fp32: no lazily gathered masters)
        if rank == 0:
            ret["loss"] = mean_loss
            ret["sd"] = {k: v.detach().cpu() for k, v in model.state_dict().items()}
        if rank == world - 1:  # the replicas end the step bit-identical (checked against rank 0's below): digests, not 0.5 GB
            import hashlib

            ret["sd_last"] = {k: hashlib.sha256(v.detach().cpu().contiguous().reshape(-1).view(torch.uint8).numpy().tobytes()).hexdigest()
                              for k, v in model.state_dict().items()}
        # a second step: the collective plan of the shape is known now, so the context views pair up too -- its count is
        # the steady state of a run (the first step's context passes ran one after the other to calibrate the plan)
        ts.step(local)
        torch.cuda.synchronize()
        ret[f"collectives_steady{rank}"] = ts.engine.collectives_last_step
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_ranks_match_single_process(hip_lib, world):
    """`world` ranks x 8/world tile pairs through the engine's SyncBatchNorm exchange, the collective recompute plan and
    the gradient reducer == the fp64 ORACLE on the full batch of 8 (not only == the product on one rank): loss, running
    statistics, updated weights.  (4 ranks: the widest rehearsal a one-GPU box allows -- at most 6 processes may share
    the card; the 8-rank case is the driver's.)"""
    from msf_wsi_amd.train import PretrainStep

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    mp.spawn(_worker, args=(world, port, ret), nprocs=world, join=True)
    # every rank issued the same number of collectives and adopted the same plan.  ResNet-18: 40 encoder + 48 head
    # BatchNorm modules, each called for 2 views, forward and backward: 352 exchanges one by one (round 3).  Heads: both
    # views in one message (96 instead of 192).  Encoders: 160 with one pass after the other (the two-stream schedule of
    # the 2-rank case), 80 with the views in lockstep (4-rank case, from the second step on; the first step's context
    # passes run singly: +20, and the plan collective: +1)
    assert len({ret[f"collectives{r}"] for r in range(world)}) == 1, dict(ret)
    assert len({ret[f"collectives_steady{r}"] for r in range(world)}) == 1, dict(ret)
    want = {2: (257, 256), 4: (197, 176)}[world]
    assert (ret["collectives0"], ret["collectives_steady0"]) == want, (ret["collectives0"], ret["collectives_steady0"], want)
    assert ("views-lockstep" in ret["plan0"]) == (world == 4), ret["plan0"]
    # gradient exchange: context_, target_ and the fuser heads' 8 per-scale buckets (projector + predictor x 4 scales)
    # (sharded, world 4: every bucket is one reduce-scatter plus -- where its length is not a multiple of 16 -- a short
    #  all-reduced tail)
    assert (ret["grad_msgs0"] == 10) if world == 2 else (10 <= ret["grad_msgs0"] <= 20), ret["grad_msgs0"]
    assert len({ret[f"plan{r}"] for r in range(world)}) == 1
    print(f"world {world}: {ret['collectives0']} engine collectives per step, plan {ret['plan0']}")

    import hashlib

    for k, v in ret["sd"].items():  # rank world-1 started from OTHER weights and buffers: bit-identical to rank 0 now
        assert hashlib.sha256(v.contiguous().reshape(-1).view(torch.uint8).numpy().tobytes()).hexdigest() == ret["sd_last"][k], k
    oc = oracle_case(CASE)
    assert abs(ret["loss"] - oc["loss64"]) <= 1e-3 * max(abs(oc["loss64"]), 1e-2), (ret["loss"], oc["loss64"])
    sd2 = ret["sd"]
    for k, v in oc["sd64"].items():
        if k.endswith("num_batches_tracked"):
            assert int(sd2[k]) == int(v) == 2
        elif "running_" in k:
            assert torch.allclose(sd2[k].double(), v, rtol=1e-3, atol=1e-5), k
    gate_updated_weights([(n, sd2[n]) for n in oc["names"]], CASE,
                         f"{world} ranks (engine SyncBN + reducer): updated weights")

    # ... and the same step on one rank: identical arithmetic up to the summation order of the statistics
    model = build_product("resnet18").cuda().train()
    ts = PretrainStep(model, lr=LR, global_batch=B, dtype=torch.float32, use_scaler=False)
    (c1, c2), (t1, t2), idx = _batch()
    loss = float(ts.step(((c1.cuda(), c2.cuda()), (t1.cuda(), t2.cuda()), idx)))
    torch.cuda.synchronize()
    assert abs(ret["loss"] - loss) <= 1e-5 * max(1.0, abs(loss)), (ret["loss"], loss)
    for k, v in model.state_dict().items():
        if "running_" in k:
            assert torch.allclose(sd2[k].double(), v.detach().cpu().double(), rtol=1e-3, atol=1e-5), k
    gate_updated_weights(list(model.named_parameters()), CASE, "1 rank: updated weights")


def _enc_worker(rank, world, port, ret):
    """ResNet-50 trunk (Bottleneck: fused two-source tails, folded BatchNorm backward) under SyncBN: the Gram-matrix
    statistics and the folded sums are all-reduced like ordinary ones"""
    from helpers import MODEL_SEED
    from msf_wsi_amd.engine import Engine
    from msf_wsi_amd.models import resnet

    torch.cuda.set_device(0)
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    try:
        x, Rs = _enc_inputs()
        n = x.shape[0] // world
        enc = _enc_model(resnet, MODEL_SEED)
        enc._engine = Engine(sync_bn=True)
        feats = enc(x[rank * n:(rank + 1) * n].cuda())
        loss = sum((f * r[rank * n:(rank + 1) * n].cuda()).sum() for f, r in zip(feats, Rs))
        loss.backward()
        torch.cuda.synchronize()
        ret[f"feat{rank}"] = [f.detach().cpu() for f in feats]
        ret[f"grad{rank}"] = {k: p.grad.detach().cpu() for k, p in enc.named_parameters() if p.grad is not None}
        ret[f"rv{rank}"] = enc.layer2[0].downsample[1].running_var.detach().cpu()
    finally:
        dist.destroy_process_group()


def _enc_inputs():
    g = torch.Generator().manual_seed(21)
    x = torch.randn(8, 3, 64, 64, generator=g)
    Rs = [torch.randn(8, d, generator=g) for d in (256, 512, 1024, 2048)]
    return x, Rs


def _enc_model(resnet, seed):
    torch.manual_seed(seed)
    enc = resnet.resnet50(zero_init_residual=False, return_features=True)
    enc.fc = torch.nn.Identity()
    for m in enc.modules():  # gates wide open: rounding-level comparison (see test_encoder_gpu)
        if isinstance(m, torch.nn.BatchNorm2d):
            m.bias.data.fill_(6.0)
    return enc.cuda().train()


def test_two_ranks_resnet50_encoder_syncbn(hip_lib):
    import numpy as np
    from helpers import MODEL_SEED
    from msf_wsi_amd.engine import Engine
    from msf_wsi_amd.models import resnet

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    mp.spawn(_enc_worker, args=(2, port, ret), nprocs=2, join=True)

    x, Rs = _enc_inputs()
    enc = _enc_model(resnet, MODEL_SEED)
    enc._engine = Engine()
    feats = enc(x.cuda())
    loss = sum((f * r.cuda()).sum() for f, r in zip(feats, Rs))
    loss.backward()
    torch.cuda.synchronize()
    for s_, f in enumerate(feats):
        two = torch.cat([ret["feat0"][s_], ret["feat1"][s_]], 0).double()
        one = f.detach().cpu().double()
        assert float((two - one).norm() / one.norm()) < 1e-5, s_
    rv = enc.layer2[0].downsample[1].running_var.detach().cpu()
    # variance through the fp32 Gram matrix with a +6 sigma mean (the open-gate trick above): E[c^2] - mean^2 keeps ~4 digits
    assert torch.allclose(ret["rv0"], rv, rtol=3e-4, atol=1e-6) and torch.allclose(ret["rv1"], rv, rtol=3e-4, atol=1e-6)
    errs = {}
    for k, p in enc.named_parameters():
        if p.grad is None or (k.endswith(".bias") and "bn" in k and "bn3" not in k):
            continue  # exactly-zero true gradients (a constant in front of conv -> BatchNorm)
        one = p.grad.detach().cpu().double()
        two = ret["grad0"][k].double() + ret["grad1"][k].double()
        errs[k] = float((two - one).norm() / (one.norm() + 1e-30))
    worst = max(errs, key=errs.get)
    # two ranks accumulate the fp32 Gram matrices / folded sums in a different order than one: with the +6 sigma
    # means of the open-gate trick that is worth 3e-4..1e-3 on the gradients (a missing or doubled term would be >1e-1)
    assert np.median(list(errs.values())) < 1e-3 and errs[worst] < 5e-3, (worst, errs[worst])


def _nce_worker(rank, world, port, ret):
    from msf_wsi_amd.dist import shard_range
    from msf_wsi_amd.train import PretrainStep

    torch.cuda.set_device(0)
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    try:
        model = build_product("resnet18").cuda().train()
        ts = PretrainStep(model, lr=LR, global_batch=B, dtype=torch.float32, use_scaler=False, sync_bn=True,
                          loss="infonce")
        (c1, c2), (t1, t2), idx = _batch()
        lo, hi = shard_range(B, world, rank)
        local = ((c1[lo:hi].cuda(), c2[lo:hi].cuda()), (t1[lo * K:hi * K].cuda(), t2[lo * K:hi * K].cuda()),
                 [idx[0][lo:hi], idx[1][lo:hi]])
        ts.step(local)
        torch.cuda.synchronize()
        ret[f"loss{rank}"] = ts.epoch_loss()
    finally:
        dist.destroy_process_group()


def test_infonce_all_gather_negatives_two_ranks(hip_lib):
    """the cross-GPU negative set: with the z rows gathered over ranks (and SyncBN), the mean loss of 2 ranks x 4 tile
    pairs equals the one of 1 rank x 8 -- every row sees the same 8 (context) / 128 (target) candidates either way.
    NOTE rows of rank r sit at columns r*rows.. of the gathered matrix, i.e. a permutation of the one-rank order for the
    target group only if samples stay contiguous, which shard_range guarantees."""
    from msf_wsi_amd.train import PretrainStep

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ret = mp.get_context("spawn").Manager().dict()
    mp.spawn(_nce_worker, args=(2, port, ret), nprocs=2, join=True)
    model = build_product("resnet18").cuda().train()
    ts = PretrainStep(model, lr=LR, global_batch=B, dtype=torch.float32, use_scaler=False, loss="infonce")
    (c1, c2), (t1, t2), idx = _batch()
    loss = float(ts.step(((c1.cuda(), c2.cuda()), (t1.cuda(), t2.cuda()), idx)))
    torch.cuda.synchronize()
    assert abs(ret["loss0"] - ret["loss1"]) < 1e-12            # epoch_loss is the all-reduced sample-weighted mean
    assert abs(ret["loss0"] - loss) <= 1e-4 * max(1.0, abs(loss)), (ret["loss0"], loss)


def _ft_worker(rank, world, port, ret):
    from msf_wsi_amd.dist import shard_range
    from msf_wsi_amd.finetune import FinetuneStep
    from test_hooknet_gpu import _build, _inputs

    torch.cuda.set_device(0)
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    try:
        model = _build().cuda().train()
        x1, x2, m1, m2 = _inputs(B=4)
        lo, hi = shard_range(4, world, rank)
        ts = FinetuneStep(model, lr=1e-3, batch_size=4, lam=0.75, dtype=torch.float32, use_scaler=False, sync_bn=True)
        # this rank's LOCAL gradient, taken right before the reducer sends it, and the averaged one right after the wait
        launch, wait, snap = ts.reducer.launch, ts.reducer.wait, {}

        def spy_launch(*a, **k):
            torch.cuda.synchronize()
            snap["local"] = ts.flats.g[0].detach().clone()
            return launch(*a, **k)

        def spy_wait():
            wait()
            torch.cuda.synchronize()
            snap["avg"] = ts.flats.g[0].detach().clone()

        ts.reducer.launch, ts.reducer.wait = spy_launch, spy_wait
        loss, _ = ts.step((x1[lo:hi].cuda(), x2[lo:hi].cuda()), (m1[lo:hi].cuda(), m2[lo:hi].cuda()))
        torch.cuda.synchronize()
        ret[f"g_local{rank}"] = snap["local"].cpu()
        ret[f"g_avg{rank}"] = snap["avg"].cpu()
        ret[f"w{rank}"] = ts.flats.w[0].detach().cpu()
        if rank == 0:
            ret["sd"] = {k: v.detach().cpu() for k, v in model.state_dict().items()}
            ret["collectives"] = ts.engine.collectives_last_step
    finally:
        dist.destroy_process_group()


def test_finetune_step_two_ranks_match_single_process(hip_lib):
    """the fused fine-tune step data-parallel (tools/ssl_finetune.py:183-193: SyncBatchNorm + DDP): 2 ranks x 2 tile pairs
    == 1 rank x 4 -- BatchNorm running statistics and updated weights.  (The Dice loss is a ratio of batch sums, so the
    per-rank LOSS VALUES differ from the full-batch one also in the reference; the gradients are what DDP averages.)"""
    from msf_wsi_amd.finetune import FinetuneStep
    from test_hooknet_gpu import _build, _inputs

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ret = mp.get_context("spawn").Manager().dict()
    mp.spawn(_ft_worker, args=(2, port, ret), nprocs=2, join=True)
    assert ret["collectives"] > 100
    # the gradient exchange: what every rank steps on is the MEAN of the two ranks' local gradients -- not a
    # local gradient (exchange missing) and not their sum -- and both ranks end with identical weights
    mean = (ret["g_local0"].double() + ret["g_local1"].double()) / 2
    assert float((ret["g_local0"] - ret["g_local1"]).abs().max()) > 0, "the shards must produce different local gradients"
    for r in range(2):
        assert float((ret[f"g_avg{r}"].double() - mean).abs().max()) <= 1e-6 * float(mean.abs().max()), r
    assert torch.equal(ret["w0"], ret["w1"])
    sd2 = ret["sd"]
    for k, v in sd2.items():
        if k.endswith("num_batches_tracked"):
            assert int(v) == 1
    # SyncBatchNorm: the running statistics of the 2-rank run are those of the full batch
    model = _build().cuda().train()
    x1, x2, m1, m2 = _inputs(B=4)
    ts = FinetuneStep(model, lr=1e-3, batch_size=4, lam=0.75, dtype=torch.float32, use_scaler=False)
    ts.step((x1.cuda(), x2.cuda()), (m1.cuda(), m2.cuda()))
    torch.cuda.synchronize()
    for k, v in model.state_dict().items():
        if "running_" in k:
            assert torch.allclose(sd2[k].double(), v.detach().cpu().double(), rtol=1e-3, atol=1e-5), k


def _shard_opt_worker(rank, world, port, ret, dtype):
    """the same two steps with the all-reduce exchange and with the sharded optimizer, on `world` ranks sharing the card"""
    import os

    os.environ["MSFWSI_TUNING"] = "15=1"  # one workgroup per weight-gradient tile: a rank's two runs repeat bit for bit
    from msf_wsi_amd.dist import shard_range
    from msf_wsi_amd.train import PretrainStep

    torch.cuda.set_device(0)
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    try:
        (c1, c2), (t1, t2), idx = _batch()
        lo, hi = shard_range(B, world, rank)
        local = ((c1[lo:hi].cuda(), c2[lo:hi].cuda()), (t1[lo * K:hi * K].cuda(), t2[lo * K:hi * K].cuda()),
                 [idx[0][lo:hi], idx[1][lo:hi]])
        for shard in (False, True):
            model = build_product("resnet18").cuda().train()
            ts = PretrainStep(model, lr=LR, global_batch=B, dtype=dtype, sync_bn=True, shard_optimizer=shard)
            assert ts.reducer.sharding == shard
            for _ in range(2):
                ts.step(local)
            torch.cuda.synchronize()
            if shard and dtype != torch.float32:
                # the fuser heads' fp32 masters are current on their owners only: state_dict refuses until they are fetched
                try:
                    model.state_dict()
                    ret[f"guard{rank}"] = False
                except RuntimeError:
                    ret[f"guard{rank}"] = True
            ck = ts.checkpoint(0)  # collective under sharding (masters + Adam moments gathered)
            # two ranks: a + b is the same sum either way, so everything is compared for EQUALITY -- digests travel to the
            # parent, not 1.5 GB of tensors through the manager's pickle (that was 55 s of the suite per case)
            import hashlib

            dg = (lambda t: hashlib.sha256(t.detach().cpu().contiguous().reshape(-1).view(torch.uint8).numpy().tobytes()).hexdigest()) \
                if world == 2 else (lambda t: t.detach().cpu())
            ret[(shard, rank)] = ({k: dg(v) for k, v in ck["state_dict"].items()},
                                  {i: {n: dg(t) for n, t in st.items() if n != "step"} for i, st in ck["optimizer"]["state"].items()
                                   if i % 37 == 0},
                                  [dg(w) for w in ts.flats.w16 if w is not None],
                                  ts.reducer.bytes_last_step, ts.t)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,dtype", [(2, torch.float32), (2, torch.bfloat16)])
def test_sharded_optimizer_equals_allreduce_step(hip_lib, world, dtype):
    """reduce-scatter + Adam on 1/world of every bucket + all-gather of the updated weights (tools/ssl_train.py:281-310,473
    as a sharded step) against the all-reduce exchange: two steps, every rank ends with the same weights, and they equal
    the unsharded run bit for bit (two ranks: a + b is the same sum either way; four ranks run the sharded step against the
    fp64 oracle in test_ranks_match_single_process[4]); 16-bit run: GradScaler's inf flag is agreed by a MAX all-reduce,
    the fuser heads travel as their 16-bit copy, state_dict refuses stale masters, checkpoint() gathers masters and moments"""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    mp.spawn(_shard_opt_worker, args=(world, port, ret, dtype), nprocs=world, join=True)
    exact = world == 2
    for r in range(world):
        sd_a, opt_a, w16_a, bytes_a, t_a = ret[(False, r)]
        sd_s, opt_s, w16_s, bytes_s, t_s = ret[(True, r)]
        assert t_a == t_s == 2
        same = (lambda a, b: a == b) if exact else torch.equal  # (two ranks: sha256 digests, see the worker)
        for k in sd_a:
            a, b = sd_a[k], sd_s[k]
            assert (same(a, b) if exact else torch.allclose(a.double(), b.double(), rtol=1e-5, atol=1e-7)), (r, k)
        for i in opt_a:
            for n in ("exp_avg", "exp_avg_sq"):
                a, b = opt_a[i][n], opt_s[i][n]
                assert (same(a, b) if exact else torch.allclose(a.double(), b.double(), rtol=1e-4, atol=1e-10)), (r, i, n)
        for a, b in zip(w16_a, w16_s):
            assert same(a, b) if exact else True
        # same weights on every rank after the sharded step
        for k in sd_s:
            assert same(sd_s[k], ret[(True, 0)][0][k]), (r, k)
        assert bytes_s <= bytes_a  # the reduce-scatter moves the buckets once; the all-reduce's second half is the all-gather
        if dtype != torch.float32:
            assert ret[f"guard{r}"] is True
