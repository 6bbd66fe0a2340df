"""CPU / gloo: the data-parallel host logic (msf_wsi_amd/dist.py) with world_size 2.

The arithmetic on each rank is done by the oracle (no GPU here); what is under test is the product's
protocol: whole-sample sharding, the packed [sum, sumsq] SUM all-reduce for cross-replica BatchNorm
(forward) and [sum g, sum g*x] (backward) with the same dx = k1*g + k2*x + k3 coefficients the HIP kernel
`bn_bwd_finalize` uses, flat per-group gradient buffers and their averaging.  Claim verified: an N-rank step
equals the single-process step on the concatenated batch."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from helpers import build_product

SIZE, B_GLOBAL, K = 32, 4, 16


class _SyncBN(torch.autograd.Function):
    """train-mode BatchNorm over all ranks, restating the engine's kernels (bn_finalize / bn_bwd_finalize)"""

    @staticmethod
    def forward(ctx, x, w, b, eps):
        from msf_wsi_amd.dist import sync_sums, world_size

        C = x.shape[1]
        xs = x.transpose(0, 1).reshape(C, -1)
        packed = torch.cat([xs.sum(1), (xs * xs).sum(1)]).double()
        sync_sums(packed)
        total = xs.shape[1] * world_size()
        mean = packed[:C] / total
        var = (packed[C:] / total - mean * mean).clamp_min(0)
        invstd = 1.0 / torch.sqrt(var + eps)
        shape = [1, C] + [1] * (x.dim() - 2)
        g = w if w is not None else torch.ones(C, dtype=x.dtype)
        bb = b if b is not None else torch.zeros(C, dtype=x.dtype)
        scale = g * invstd
        ctx.save_for_backward(x, mean, invstd, g)
        ctx.total, ctx.affine = total, w is not None
        return x * scale.view(shape) + (bb - mean * scale).view(shape)

    @staticmethod
    def backward(ctx, dy):
        from msf_wsi_amd.dist import sync_sums

        x, mean, invstd, g = ctx.saved_tensors
        C = x.shape[1]
        shape = [1, C] + [1] * (x.dim() - 2)
        dys = dy.transpose(0, 1).reshape(C, -1)
        xs = x.transpose(0, 1).reshape(C, -1)
        local = torch.cat([dys.sum(1), (dys * xs).sum(1)]).double()
        dgamma = invstd * (local[C:] - mean * local[:C])  # LOCAL sums: DDP averages parameter grads later
        dbeta = local[:C].clone()
        glob = sync_sums(local.clone())
        dot = invstd * (glob[C:] - mean * glob[:C])
        a = g * invstd
        m1, m2 = glob[:C] / ctx.total, dot / ctx.total
        k1, k2, k3 = a, -a * m2 * invstd, -a * m1 + a * m2 * invstd * mean
        dx = k1.view(shape) * dy + k2.view(shape) * x + k3.view(shape)
        return dx, (dgamma if ctx.affine else None), (dbeta if ctx.affine else None), None


def _grads(sd, batch, sync):
    from oracle import msfwsi_oracle as orc

    keep = orc._bn
    if sync:
        orc._bn = lambda s, key, x, train=True: _SyncBN.apply(x, s.get(key + ".weight"), s.get(key + ".bias"),
                                                              orc.BN_EPS)
    try:
        opt = orc.Adam(sd, [0.0, 0.0, 0.0])
        opt.step = lambda *a, **k: None
        loss, terms, outs, grads = orc.train_step(sd, batch, opt)
    finally:
        orc._bn = keep
    return loss, grads


def _worker(rank, world, port, ret):
    from msf_wsi_amd.dist import FlatGroups, GradReducer, shard_range
    from oracle import msfwsi_oracle as orc

    torch.set_num_threads(2)
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    try:
        model = build_product("resnet18").double()
        sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
        (c1, c2), (t1, t2), idx = orc.synthetic_batch(B_GLOBAL, SIZE, K, 0, torch.float64)
        lo, hi = shard_range(B_GLOBAL, world, rank)
        local = ((c1[lo:hi], c2[lo:hi]), (t1[lo * K:hi * K], t2[lo * K:hi * K]), [idx[0][lo:hi], idx[1][lo:hi]])
        loss, grads = _grads({k: v.clone() for k, v in sd.items()}, local, sync=True)
        # product plumbing: flat per-group gradient buffers + per-group averaging
        for p in model.parameters():
            p.data = p.data.float()
        flats = FlatGroups(model, with_bf16=False, device="cpu")
        for n, p in model.named_parameters():
            gv = flats.grad_view(p)
            gsrc = grads[n].float()
            gv.copy_(gsrc.permute(0, 2, 3, 1) if gsrc.dim() == 4 else gsrc)
        red = GradReducer(flats)
        # the order the backward schedule releases them (Engine.model_backward): the fuser heads per scale from the widest
        # down -- predictor bucket, projector bucket -- then whatever of inter_ no bucket covered, then target_, context_
        for s_ in (3, 2, 1, 0):
            red.launch("inter", part=f"inter_predictor.{s_}.")
            red.launch("inter", part=f"inter_projector.{s_}.")
        for name in ("inter", "target", "context"):
            red.launch(name)
        red.wait()
        if rank == 0:
            ret["grad_msgs"] = red.launches_last_step
            ret["grad_bytes"] = red.bytes_last_step
        lsum = torch.tensor([float(loss)], dtype=torch.float64)
        dist.all_reduce(lsum)
        if rank == 0:
            full_loss, full = _grads({k: v.clone() for k, v in sd.items()},
                                     ((c1, c2), (t1, t2), idx), sync=False)
            worst = 0.0
            for n, p in model.named_parameters():
                gv = flats.grad_view(p)
                got = gv.permute(0, 3, 1, 2) if p.dim() == 4 else gv
                ref = full[n].float()
                worst = max(worst, float((got - ref).norm() / (ref.norm() + 1e-30)))
            ret["worst"] = worst
            ret["loss"] = (float(lsum) / world, float(full_loss))
    finally:
        dist.destroy_process_group()


def _pair_worker(rank, world, port, ret):
    """two 'view' threads per rank meet at every exchange (engine._ViewPair); the second to arrive issues ONE all-reduce
    over both views' rows"""
    import threading

    from msf_wsi_amd.dist import sync_sums
    from msf_wsi_amd.engine import _ViewPair

    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    try:
        pair, n, rounds = _ViewPair(), 6, 7
        both = torch.zeros(2, n, dtype=torch.float64)
        calls, got, order = [], {0: [], 1: []}, []

        def collective():
            calls.append(threading.current_thread().name)
            sync_sums(both.view(-1))

        def view(v):
            for r in range(rounds):
                both[v] = float(100 * rank + 10 * v + r)          # this view's packed statistics of exchange r
                pair.exchange(v, collective)
                got[v].append(both[v].clone())
                pair.turn_wait(v)                                  # view 0's running-statistics update goes first
                order.append((r, v))
                pair.turn_done(v)

        t = threading.Thread(target=view, args=(1,), name="view1")
        t.start()
        view(0)
        t.join()
        want = lambda v, r: sum(100 * k + 10 * v + r for k in range(world))
        ok = all(float(got[v][r][0]) == want(v, r) for v in (0, 1) for r in range(rounds))
        turn_ok = all(order.index((r, 0)) < order.index((r, 1)) for r in range(rounds))
        ret[rank] = (len(calls), ok, turn_ok)
    finally:
        dist.destroy_process_group()


def test_view_pair_one_collective_per_exchange():
    """the lockstep rendezvous of the two views (more than one rank: one SyncBatchNorm message per BatchNorm for both views)
    on CPU tensors over gloo: one collective per exchange whichever view arrives last, each view reads its own row of
    the summed message, view 0 passes every turn point before view 1; a failing view fails its partner instead of
    leaving it waiting"""
    import threading

    from msf_wsi_amd.engine import _ViewPair

    ret = mp.Manager().dict()
    mp.spawn(_pair_worker, args=(2, _free_port(), ret), nprocs=2, join=True)
    assert ret[0] == (7, True, True) and ret[1] == (7, True, True), dict(ret)
    pair, seen = _ViewPair(), []

    def partner():
        try:
            pair.exchange(1, lambda: None)
            pair.exchange(1, lambda: None)  # view 0 has failed by now: this must raise, not hang
        except RuntimeError as e:
            seen.append(str(e))

    t = threading.Thread(target=partner)
    t.start()
    pair.exchange(0, lambda: None)
    pair.fail(ValueError("view 0 broke"))
    t.join(timeout=30)
    assert not t.is_alive() and seen and "other view pass" in seen[0]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def test_two_ranks_equal_full_batch():
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(2, _free_port(), ret), nprocs=2, join=True)
    assert abs(ret["loss"][0] - ret["loss"][1]) < 1e-9, ret["loss"]
    assert ret["grad_msgs"] == 10, ret["grad_msgs"]  # 8 per-scale buckets of inter_ + target_ + context_
    assert ret["worst"] < 1e-5, ret["worst"]  # grads travelled as fp32 through the flat buffers


def test_shard_range_and_layout():
    from msf_wsi_amd.dist import ALIGN, FlatGroups, shard_range

    assert shard_range(2048, 8, 3) == (768, 1024)
    with pytest.raises(ValueError):
        shard_range(10, 4, 0)
    model = build_product("resnet18")
    before = {n: p.detach().clone() for n, p in model.named_parameters()}
    flats = FlatGroups(model, with_bf16=True, device="cpu")
    assert [len(g) for g in flats.params] == [108, 108, 48]
    for gi in range(3):
        assert all(o % ALIGN == 0 for o in flats.offsets[gi]) and flats.sizes[gi] % ALIGN == 0
        assert flats.w16[gi].dtype == torch.bfloat16 and flats.w16[gi].numel() == flats.sizes[gi]
    for n, p in model.named_parameters():
        assert torch.equal(p.detach(), before[n]), n  # values and logical shapes unchanged
        gi = [i for i, names in enumerate(flats.names) if n in names][0]
        lo = flats.w[gi].data_ptr()
        assert lo <= p.data_ptr() < lo + flats.w[gi].numel() * 4  # the parameter IS a view of the flat buffer
        if p.dim() == 4:
            assert p.permute(0, 2, 3, 1).is_contiguous()
        assert flats.grad_view(p).numel() == p.numel() and flats.grad_view(p).data_ptr() % 16 == 0
    # load_state_dict writes through the views
    sd = {k: torch.zeros_like(v) for k, v in model.state_dict().items()}
    model.load_state_dict(sd)
    assert all(float(w.abs().sum()) == 0 for w in flats.w)


def test_bench_self_launcher_rendezvous():
    """`python bench.py --gpus 2` as the driver may call it (no WORLD_SIZE in the environment): the parent starts one
    rank per GPU through torch.distributed.run over 127.0.0.1, the ranks form the process group, pass the collective
    capability probe and a barrier, rank 0 prints one JSON line and the parent returns the ranks' exit code
    (gloo here: there is no GPU in this container; the compute part needs an MI355X)"""
    import json
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["MSFWSI_BENCH_BACKEND"] = "gloo"
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--rendezvous-only"], env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    assert json.loads(line) == {"rendezvous": "ok", "world": 2, "backend": "gloo"}
    # without a GPU the compute path refuses loudly (no CPU fallback), and the launcher passes the failure on
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1"], env=env,
                       capture_output=True, text=True, timeout=600)
    if not __import__("torch").cuda.is_available():
        assert r.returncode != 0


def _plan_worker(rank, world, port, ret):
    """two ranks whose LOCAL measurements differ (bytes per image, free memory -> local proposal) must issue the plan
    collective on the same steps and adopt the same, most conservative, plan (the cache key used to hold
    the locally measured bytes per image, so one rank could skip the collective while its peer issued it)"""
    from msf_wsi_amd.engine import Engine

    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    try:
        eng = Engine(sync_bn=True)
        eng.recompute = "auto"
        local = [(set(), False), ({"t1"}, False)][rank]        # rank 1 is short of memory
        eng._plan_local = lambda *a, **k: (set(local[0]), local[1], True)
        got = []
        for step in range(3):
            per_image = 22.5e6 + 4096 * rank + 512 * step       # allocator noise: differs per rank AND per step
            nosave = eng._plan_recompute(per_image, 8, 16, torch.device("cpu"), 0.0, shape_key=((3, 64, 64), "r18"))
            got.append((tuple(sorted(nosave)), eng.last_plan, eng.collectives))
        nosave = eng._plan_recompute(1e6, 4, 16, torch.device("cpu"), 0.0, shape_key=((3, 64, 64), "r18"))  # a new shape
        got.append((tuple(sorted(nosave)), eng.last_plan, eng.collectives))
        ret[rank] = got
    finally:
        dist.destroy_process_group()


def test_collective_recompute_plan_is_rank_invariant():
    ret = mp.Manager().dict()
    mp.spawn(_plan_worker, args=(2, _free_port(), ret), nprocs=2, join=True)
    assert ret[0] == ret[1], (ret[0], ret[1])
    assert [g[0] for g in ret[0]] == [("t1",)] * 4 and ret[0][0][1].startswith("recompute:t1")
    assert [g[2] for g in ret[0]] == [1, 1, 1, 2]  # one plan collective per shape, on both ranks alike


def test_shard_bucket_partitions_every_bucket():
    """the sharded optimizer's split of a gradient bucket (dist.shard_bucket): the ranks' shards and the common tail tile
    [lo, hi) exactly once, shards start on 16-byte groups of fp32 (the Adam kernel's float4 steps), equal shard lengths
    (reduce-scatter needs them), tail shorter than world * (align + 1)"""
    from msf_wsi_amd.dist import shard_bucket

    for world in (1, 2, 3, 4, 8):
        for lo, n in ((0, 1), (64, 7), (128, 64), (0, 4097), (256, 18432 * 18432 // 64 + 5), (64, 1000003)):
            hi = lo + n
            cover = []
            pers = set()
            for r in range(world):
                per, own, tail = shard_bucket(lo, hi, world, r)
                pers.add(per)
                assert per % 4 == 0 and own[1] - own[0] == per and (own[0] - lo) % 4 == 0
                assert tail[1] == hi and tail[0] == lo + per * world and (tail[0] - lo) % 4 == 0
                assert tail[1] - tail[0] < world * 5
                if per:
                    cover.append(own)
            assert len(pers) == 1
            cover.sort()
            pos = lo
            for a, b in cover:
                assert a == pos
                pos = b
            assert pos == tail[0]


def _shard_worker(rank, world, port, ret):
    """the reducer's sharded exchange on CPU tensors over gloo: every rank ends with the MEAN gradient on the ranges it owns,
    a stand-in update w -= g on those ranges followed by gather_weights leaves every rank with the weights of the unsharded
    step, bit for bit"""
    from msf_wsi_amd.dist import FlatGroups, GradReducer

    torch.set_num_threads(2)
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    try:
        torch.manual_seed(5)
        model = build_product("resnet18")
        flats = FlatGroups(model, with_bf16=False, device="cpu")
        w0 = [w.clone() for w in flats.w]
        gens = [torch.Generator().manual_seed(100 + r) for r in range(world)]
        local = [[torch.randn(g.numel(), generator=gens[r]) for g in flats.g] for r in range(world)]  # every rank's gradients
        mean = [sum(local[r][gi] for r in range(world)) / world for gi in range(len(flats.g))]
        for gi, g in enumerate(flats.g):
            g.copy_(local[rank][gi])
        red = GradReducer(flats, None, shard=True)
        assert red.sharding
        # the trainer's order: the fuser heads' per-scale buckets, then what is left of the group, then the other groups
        for part in ("inter_projector.3.", "inter_predictor.3.", "inter_projector.0."):
            red.launch("inter", part=part)
        red.launch("inter")
        red.launch("target")
        red.launch("context")
        red.wait()
        owned, scattered = red.take_shards()
        covered = 0
        for gi in range(len(flats.g)):
            for lo, hi in owned.get(gi, []):
                assert torch.equal(flats.g[gi][lo:hi], mean[gi][lo:hi]) or torch.allclose(flats.g[gi][lo:hi], mean[gi][lo:hi], rtol=0, atol=1e-6)
                flats.w[gi][lo:hi] -= flats.g[gi][lo:hi]
                covered += hi - lo
        for gi in range(len(flats.g)):
            for wk in red.gather_weights(gi, scattered.get(gi, []), flats.w[gi]):
                wk.wait()
        ret[f"w{rank}"] = [w.clone() for w in flats.w]
        ret[f"covered{rank}"] = covered
        if rank == 0:
            ret["ref"] = [w0[gi] - mean[gi] for gi in range(len(w0))]
            ret["total"] = sum(g.numel() for g in flats.g)
            ret["msgs"] = red.launches_last_step
    finally:
        dist.destroy_process_group()


def test_sharded_reducer_equals_allreduce_step():
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    world = 2
    mp.spawn(_shard_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    for gi, ref in enumerate(ret["ref"]):
        for r in range(world):
            assert torch.allclose(ret[f"w{r}"][gi], ref, rtol=0, atol=1e-6), (gi, r)
        assert torch.equal(ret["w0"][gi], ret["w1"][gi])  # identical weights on every rank
    # each rank stepped about half of the parameters (its shards + the short tails of 6 buckets)
    assert ret["total"] // 2 <= ret["covered0"] <= ret["total"] // 2 + 6 * world * 5


def _bcast_worker(rank, world, port, ret):
    """the DDP constructor's rank-0 broadcast (tools/ssl_train.py:170) through dist.broadcast_state: every rank builds its
    model from a DIFFERENT seed (the reference seeds the parent process only, :46-48; mp.spawn workers draw their own), lays
    it out in the flat groups, and the broadcast leaves weights, views and BatchNorm buffers bit-identical to rank 0's"""
    from msf_wsi_amd.dist import FlatGroups, broadcast_state, probe_sharded

    torch.set_num_threads(2)
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    try:
        model = build_product("resnet18")
        with torch.no_grad():  # rank-dependent weights AND buffers
            g = torch.Generator().manual_seed(100 + rank)
            for p in model.parameters():
                p.add_(torch.randn(p.shape, generator=g))
            for b in model.buffers():
                b.add_(rank + 1)
        flats = FlatGroups(model, with_bf16=False, device="cpu")
        before = float(model.inter_projector[3][0].weight.double().sum())
        broadcast_state(list(flats.w) + list(model.buffers()))
        sd = model.state_dict()
        ret[rank] = {"before": before, "sum": {k: float(v.double().sum()) for k, v in sd.items()},
                     "flat": [float(w.double().abs().sum()) for w in flats.w],
                     "sharded_ok": probe_sharded(None, "cpu")}
    finally:
        dist.destroy_process_group()


def test_rank0_broadcast_makes_differently_seeded_replicas_equal():
    ret = mp.Manager().dict()
    mp.spawn(_bcast_worker, args=(2, _free_port(), ret), nprocs=2, join=True)
    assert ret[0]["before"] != ret[1]["before"]          # the replicas really differed
    assert ret[0]["sum"] == ret[1]["sum"] and ret[0]["flat"] == ret[1]["flat"]   # bit-identical sums of every entry
    assert len(ret[0]["sum"]) == 528
    assert ret[0]["sum"]["context_encoder.bn1.num_batches_tracked"] == 1.0       # rank 0's buffers (0 + 1), not rank 1's
    # gloo runs the in-place reduce-scatter / all-gather forms (or both ranks agree that it does not): same branch everywhere
    assert ret[0]["sharded_ok"] == ret[1]["sharded_ok"]


def _world8_worker(rank, world, port, ret):
    """what the driver's 8-rank run does before its first kernel, on CPU tensors over gloo: rendezvous, the collective
    capability probes, the rank-0 broadcast, the collective recompute plan with one rank short of memory, and one sharded
    gradient exchange + gather over the trainer's bucket order"""
    from msf_wsi_amd.dist import FlatGroups, GradReducer, broadcast_state, probe_collectives, probe_sharded, shard_bucket, shard_range
    from msf_wsi_amd.engine import Engine

    torch.set_num_threads(1)
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    try:
        probe_collectives(None, "cpu")
        sharded = probe_sharded(None, "cpu")
        torch.manual_seed(1000 + rank)
        model = torch.nn.ModuleDict({"context_a": torch.nn.Linear(70, 33), "target_a": torch.nn.Conv2d(8, 16, 3),
                                     "inter_projector": torch.nn.ModuleList([torch.nn.Linear(130, 130, bias=False) for _ in range(4)]),
                                     "inter_predictor": torch.nn.ModuleList([torch.nn.Linear(130, 32) for _ in range(4)])})
        flats = FlatGroups(model, with_bf16=False, device="cpu")
        broadcast_state(flats.w)
        w0 = [w.clone() for w in flats.w]
        # collective plan: rank 5 is short of memory -> every rank recomputes t1, one collective per shape
        eng = Engine(sync_bn=True)
        eng.recompute = "auto"
        eng._plan_local = lambda *a, **k: (({"t1"}, False, False) if rank == 5 else (set(), False, True))
        nosave = eng._plan_recompute(22.5e6 + rank, 256, 16, torch.device("cpu"), 0.0, shape_key=((3, 224, 224), "r50"))
        gens = [torch.Generator().manual_seed(7 + r) for r in range(world)]
        local = [[torch.randn(g.numel(), generator=gens[r]) for g in flats.g] for r in range(world)]
        mean = [sum(local[r][gi] for r in range(world)) / world for gi in range(len(flats.g))]
        for gi, g in enumerate(flats.g):
            g.copy_(local[rank][gi])
        red = GradReducer(flats, None, shard=sharded)
        for s_ in (3, 2, 1, 0):
            red.launch("inter", part=f"inter_predictor.{s_}.")
            red.launch("inter", part=f"inter_projector.{s_}.")
        for name in ("inter", "target", "context"):
            red.launch(name)
        red.wait()
        owned, scattered = red.take_shards()
        own_elems = 0
        for gi in range(len(flats.g)):
            for lo, hi in (owned.get(gi, []) if sharded else [(0, flats.g[gi].numel())]):
                assert torch.allclose(flats.g[gi][lo:hi], mean[gi][lo:hi], rtol=0, atol=1e-6)
                flats.w[gi][lo:hi] -= flats.g[gi][lo:hi]
                own_elems += hi - lo
            for lo, per in scattered.get(gi, []):  # the ownership rule the Adam pass relies on
                assert (lo + rank * per, lo + (rank + 1) * per) in owned[gi]
        for gi in range(len(flats.g)):
            for wk in red.gather_weights(gi, scattered.get(gi, []), flats.w[gi]):
                wk.wait()
        ret[rank] = {"plan": (tuple(sorted(nosave)), eng.last_plan, eng.collectives), "sharded": sharded,
                     "shard": shard_range(2048, world, rank), "own": own_elems,
                     "w": [w.clone() for w in flats.w], "ref": [w0[gi] - mean[gi] for gi in range(len(w0))],
                     "msgs": red.launches_last_step}
    finally:
        dist.destroy_process_group()


def test_eight_ranks_rendezvous_plan_and_shard_ownership():
    """world 8 is what the driver's scaling run uses (BASELINE config 3: 8 x 256 tile pairs): eight gloo processes on the
    CPU form the group, agree on the collective probes and the recompute plan, and the sharded exchange over the trainer's
    bucket order leaves every rank with the weights of the unsharded step"""
    world = 8
    ret = mp.Manager().dict()
    mp.spawn(_world8_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    assert len(ret) == world
    assert len({ret[r]["plan"] for r in range(world)}) == 1 and ret[0]["plan"][0] == ("t1",) and ret[0]["plan"][2] == 1
    assert len({ret[r]["sharded"] for r in range(world)}) == 1
    assert [ret[r]["shard"] for r in range(world)] == [(256 * r, 256 * (r + 1)) for r in range(world)]
    assert len({ret[r]["msgs"] for r in range(world)}) == 1
    total = sum(w.numel() for w in ret[0]["w"])
    if ret[0]["sharded"]:
        assert all(ret[r]["own"] < total // 4 for r in range(world))        # each rank stepped ~1/8 (+ tails)
    for r in range(world):
        for gi, ref in enumerate(ret[0]["ref"]):
            assert torch.allclose(ret[r]["w"][gi], ref, rtol=0, atol=1e-6), (r, gi)
            assert torch.equal(ret[r]["w"][gi], ret[0]["w"][gi])
