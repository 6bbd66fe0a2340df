"""The MSF-WSI pre-train step as one fused, graph-free schedule on MI355X.

`PretrainStep.step(batch)` performs exactly one iteration of the reference's hot loop
(tools/ssl_train.py:425-474): forward of MSFWSI (src/models/backbone.py:129-222), the weighted
negative-cosine loss over 3 groups x 4 scales (:448-466), backward, gradient averaging across data-parallel
ranks (DDP, :170), GradScaler protocol (:100, :472-474) and the 3-group Adam update (:281-310) -- with every
arithmetic operation in the hand-written gfx950 kernels, no host synchronisation inside the step (the
reference's per-step `loss.item()`, :467, is replaced by a device-side accumulator), flat parameter groups
(one Adam launch and one RCCL all-reduce per group) and the heads' gradients in flight over xGMI while the
encoders are still in backward.

Checkpoints keep the reference's dict layout (:375-386): {"epoch","arch","state_dict" (keys prefixed
"module."),"optimizer" (torch.optim.Adam format, 3 groups),"scaler" (GradScaler format)}; `resume` reproduces
the reference's quirk of forcing eps=0.1 afterwards (:325-326).
"""
from __future__ import annotations

import math
import os
from typing import Dict, Optional, Sequence, Tuple

import torch
import torch.nn as nn

from . import _lib
from . import kernels as kn
from .dist import FlatGroups, GradReducer, broadcast_state, probe_sharded, world_size
from .engine import Engine, GradStore, WeightStore

FUSER_WEIGHTS = (0.1, 0.4, 0.7, 1.0)  # tools/ssl_train.py:623-625


class _FlatGradStore(GradStore):
    """gradient accumulators that live inside the flat per-group buffers"""

    preallocated = True  # every accumulator exists (and is zeroed) before the backward starts: two streams may add

    def __init__(self, flats: FlatGroups):
        super().__init__()
        self.flats = flats
        self._marked: set = set()

    def get(self, param):
        return self.flats.grad_view(param)

    def logical(self, param):
        b = self.flats.grad_view(param)
        return b.permute(0, 3, 1, 2) if param.dim() == 4 else b

    def mark_stored(self, param):
        """the engine writes this gradient with a storing launch every step (Engine.chain_backward_pair)"""
        if id(param) not in self._marked:
            self._marked.add(id(param))
            self.flats.mark_stored(param)

    def unmark_stored(self, param):
        """this step ACCUMULATES into the gradient (more rows than the storing launch takes, or the knob was flipped): a
        tensor stored on an earlier step still holds that step's values -- clear it and hand it back to zero_grads"""
        if id(param) in self._marked:
            self._marked.discard(id(param))
            self.flats.unmark_stored(param)


class FlatAdamScaler:
    """The optimizer half of a fused step, shared by PretrainStep and finetune.FinetuneStep: flat per-group fp32
    master weights / gradients / Adam moments (dist.FlatGroups), Adam's step count and the GradScaler state resident on
    the device, one Adam launch per group, state dicts in torch.optim.Adam / torch.amp.GradScaler format.
    Subclasses set: flats, lrs, eps, betas, dtype, device, engine, init_lr."""

    def _init_optimizer_state(self, use_scaler: Optional[bool], init_scale: float):
        dev, dtype = self.device, self.dtype
        # Adam's step count lives on the device: it advances only on steps the GradScaler does not skip
        # (scaler.step(optimizer), ssl_train.py:473), and the kernel forms the bias corrections from it
        self.step_dev = torch.zeros(1, dtype=torch.int32, device=dev)
        # GradScaler state, resident on the device (the reference enables it whenever --amp, also for bf16)
        self.use_scaler = (dtype != torch.float32) if use_scaler is None else bool(use_scaler)
        self.scale = torch.full((1,), init_scale if self.use_scaler else 1.0, dtype=torch.float32, device=dev)
        self.growth_tracker = torch.zeros(1, dtype=torch.int32, device=dev)
        self.found_inf = torch.zeros(1, dtype=torch.float32, device=dev)
        self.growth_factor, self.backoff_factor, self.growth_interval = 2.0, 0.5, 2000
        self.loss_accum = torch.zeros(1, dtype=torch.float64, device=dev)

    def _register_lowp_weights(self):
        """16-bit compute copies live in the flat buffers and are refreshed by the Adam kernel in the same pass"""
        if self.dtype == torch.float32:
            return
        for plist in self.flats.params:
            for p in plist:
                if p.dim() >= 2 and not (p.dim() == 4 and p.shape[1] == 3):  # stem: padded copy, cached
                    self.engine.weights.register(p, self.dtype, self.flats.w16_view(p))
        for gi in range(len(self.flats.w)):
            kn.cast_lowp(self.flats.w[gi], self.flats.w16[gi])


    @property
    def t(self) -> int:
        """Adam's step count (host read-back: synchronises; used by checkpointing and tests only)"""
        return int(self.step_dev.item())

    # optimizer groups whose fp32 master weights no kernel of a step reads (only Adam and state_dict do): after a SHARDED
    # step their 16-bit compute copy alone is all-gathered; the masters of the other ranks' shards are fetched on demand
    # (sync_master_weights: checkpoints, state_dict).  PretrainStep: the `inter_` heads, 95 % of the parameters.
    lazy_master_groups: Tuple[int, ...] = ()
    _master_stale = False
    _last_scattered: Optional[dict] = None

    def optimizer_step(self):
        """GradScaler's inf check + unscale, Adam, scaler update (ssl_train.py:473-474).  With a sharding reducer
        (dist.GradReducer(shard=True)): the check and Adam run on this rank's shard of every bucket, then the updated
        weights are all-gathered -- the fp32 masters where kernels read them, the 16-bit copy alone for lazy_master_groups."""
        reducer = getattr(self, "reducer", None)
        sharded = reducer is not None and reducer.sharding
        owned = scattered = None
        if sharded:
            owned, scattered = reducer.take_shards()
            if self._master_stale and self._last_scattered is not None and any(
                    scattered.get(gi) != self._last_scattered.get(gi) for gi in self.lazy_master_groups):
                # the fp32 masters of a lazy group are current only on the rank that stepped them LAST step; this step's
                # buckets cut the group differently, so the new owners would step stale masters: fetch them first
                # (collective; every rank sees the same layouts, so every rank takes this branch)
                self.sync_master_weights()
        ngroups = len(self.flats.w)

        def ranges(gi):  # element ranges of group gi this rank steps
            return owned.get(gi, []) if sharded else [(0, self.flats.w[gi].numel())]

        found = None
        ls = None
        if self.use_scaler:
            self.found_inf.zero_()
            for gi in range(ngroups):
                for lo, hi in ranges(gi):
                    kn.nonfinite_check(self.flats.g[gi][lo:hi], self.found_inf)
            if sharded:  # every rank must take the same skip / step decision
                import torch.distributed as dist

                dist.all_reduce(self.found_inf, op=dist.ReduceOp.MAX, group=reducer.group)
            found, ls = self.found_inf, self.scale
        kn.adam_step_advance(self.step_dev, found)
        for gi in range(ngroups):
            w16 = self.flats.w16[gi]
            for lo, hi in ranges(gi):
                if hi > lo:
                    kn.adam(self.flats.w[gi][lo:hi], self.flats.g[gi][lo:hi], self.flats.m[gi][lo:hi], self.flats.v[gi][lo:hi],
                            self.lrs[gi], self.betas[0], self.betas[1], self.eps[gi], self.step_dev, loss_scale=ls,
                            found=found, p_lowp=w16[lo:hi] if w16 is not None else None)
        if sharded:
            works = []
            for gi in range(ngroups):
                sc = scattered.get(gi, [])
                w16 = self.flats.w16[gi]
                lazy = gi in self.lazy_master_groups and w16 is not None
                if w16 is not None:
                    works += reducer.gather_weights(gi, sc, w16)
                if not lazy:
                    works += reducer.gather_weights(gi, sc, self.flats.w[gi])
            for wk in works:
                wk.wait()
            self._gather_small_masters(reducer, owned, scattered)
            self._last_scattered = scattered
            self._master_stale = any(gi in self.lazy_master_groups and self.flats.w16[gi] is not None for gi in range(ngroups))
        if self.use_scaler:
            kn.scaler_update(self.scale, self.growth_tracker, self.found_inf, self.growth_factor,
                             self.backoff_factor, self.growth_interval)
        # padded / cast copies (and the stem's filter-row runs) keyed on torch's version counter do not see
        # raw-pointer updates
        self.engine.invalidate_weights()

    _small_cache = None
    _small_layout = None

    def _gather_small_masters(self, reducer, owned, scattered):
        """lazy_master_groups keep the other ranks' fp32 masters stale -- but the 1-D parameters of those groups (BatchNorm
        weight / bias, Linear bias) have no 16-bit copy: kernels read their fp32 values directly.  They are a few thousand
        elements: every rank contributes the ones it stepped (its shards; bucket tails, which every rank steps, count on rank
        0), zeros elsewhere, and ONE all-reduce(SUM) completes the vector everywhere."""
        import torch.distributed as dist

        from .dist import rank as _rank

        lazy = [gi for gi in self.lazy_master_groups if self.flats.w16[gi] is not None]
        if not lazy:
            return
        # the ownership mask is valid for ONE bucket layout: a step whose launch(part=...) sequence differs (bucket_inter
        # toggled, another set of parts) moves the shard boundaries, and with them who holds the current master of an element
        layout = tuple((gi, tuple(scattered.get(gi, [])), tuple(owned.get(gi, []))) for gi in lazy)
        if self._small_cache is not None and self._small_layout != layout:
            self._small_cache = None
        if self._small_cache is None:
            self._small_layout = layout
            r = _rank(reducer.group)
            cache = []
            for gi in lazy:
                spans = [(off, off + p.numel()) for p, off in zip(self.flats.params[gi], self.flats.offsets[gi]) if p.dim() < 2]
                if not spans:
                    continue
                dev = self.flats.w[gi].device
                idx = torch.cat([torch.arange(a, b, device=dev) for a, b in spans])
                mine = torch.zeros_like(idx, dtype=torch.bool)
                shards = [(lo + r * per, lo + (r + 1) * per) for lo, per in scattered.get(gi, [])]
                for lo, hi in owned.get(gi, []):
                    if (lo, hi) in shards or r == 0:  # a tail (stepped identically by every rank) counts once
                        mine |= (idx >= lo) & (idx < hi)
                cache.append((gi, idx, mine.to(torch.float32)))
            self._small_cache = cache
        for gi, idx, mine in self._small_cache:
            vals = self.flats.w[gi][idx] * mine
            dist.all_reduce(vals, op=dist.ReduceOp.SUM, group=reducer.group)
            self.flats.w[gi][idx] = vals

    def sync_master_weights(self):
        """after sharded steps: fetch the other ranks' shards of the fp32 master weights of lazy_master_groups (their Adam
        moments stay sharded: optimizer_state_dict gathers them).  COLLECTIVE: every rank must call it (checkpoint, resume and
        the state_dict hook do)."""
        if not self._master_stale:
            return
        for gi in self.lazy_master_groups:
            for wk in self.reducer.gather_weights(gi, self._last_scattered.get(gi, []), self.flats.w[gi]):
                wk.wait()
        self._master_stale = False

    # ---------------------------------------------------------------------------------------
    # checkpoint interop (reference dict layout)
    # ---------------------------------------------------------------------------------------
    def _torch_adam(self) -> torch.optim.Adam:
        groups = [{"params": plist, "lr": lr, "eps": eps} for plist, lr, eps in zip(self.flats.params, self.lrs, self.eps)]
        return torch.optim.Adam(groups, lr=self.init_lr)

    def optimizer_state_dict(self) -> dict:
        reducer = getattr(self, "reducer", None)
        if reducer is not None and reducer.sharding and self._last_scattered is not None:
            # sharded steps leave each rank with the moments of its own shards: gather them (collective, like checkpoint())
            for gi in range(len(self.flats.w)):
                for buf in (self.flats.m[gi], self.flats.v[gi]):
                    for wk in reducer.gather_weights(gi, self._last_scattered.get(gi, []), buf):
                        wk.wait()
        opt = self._torch_adam()
        t = self.t
        if t > 0:
            for gi, plist in enumerate(self.flats.params):
                for pi, p in enumerate(plist):
                    m, v = self.flats.state_views(gi, pi)
                    opt.state[p] = {"step": torch.tensor(float(t)), "exp_avg": m.clone(), "exp_avg_sq": v.clone()}
        return opt.state_dict()

    def load_optimizer_state_dict(self, sd: dict):
        opt = self._torch_adam()
        opt.load_state_dict(sd)
        steps = set()
        for gi, plist in enumerate(self.flats.params):
            grp = opt.param_groups[gi]
            self.lrs[gi], self.eps[gi] = float(grp["lr"]), float(grp["eps"])
            for pi, p in enumerate(plist):
                st = opt.state.get(p)
                if not st:
                    continue
                m, v = self.flats.state_views(gi, pi)
                m.copy_(st["exp_avg"])
                v.copy_(st["exp_avg_sq"])
                steps.add(int(float(st["step"])))
        if len(steps) > 1:
            raise ValueError("per-parameter Adam step counts differ; not a checkpoint of this training loop")
        self.step_dev.fill_(steps.pop() if steps else 0)

    def scaler_state_dict(self) -> dict:
        if not self.use_scaler:
            return {}
        return {"scale": float(self.scale.item()), "growth_factor": self.growth_factor,
                "backoff_factor": self.backoff_factor, "growth_interval": self.growth_interval,
                "_growth_tracker": int(self.growth_tracker.item())}

    def load_scaler_state_dict(self, sd: dict):
        if not sd:
            return
        self.scale.fill_(float(sd["scale"]))
        self.growth_factor, self.backoff_factor = float(sd["growth_factor"]), float(sd["backoff_factor"])
        self.growth_interval = int(sd["growth_interval"])
        self.growth_tracker.fill_(int(sd["_growth_tracker"]))


class PretrainStep(FlatAdamScaler):
    def __init__(self, model: nn.Module, lr: float = 1e-3, global_batch: int = 32, ms_lr: Sequence[float] = (1, 1, 1),
                 fuser_weights: Sequence[float] = FUSER_WEIGHTS, dtype: torch.dtype = torch.bfloat16,
                 use_scaler: Optional[bool] = None, init_scale: float = 65536.0, process_group=None,
                 sync_bn: bool = True, arch: str = "resnet18", loss: str = "cosine", temperature: float = 0.2,
                 shard_optimizer: Optional[bool] = None, broadcast_from_rank0: bool = True):
        """shard_optimizer (None: on with more than one rank unless MSFWSI_SHARD_OPT=0 or the backend fails the start-up probe
        of the in-place reduce-scatter / all-gather forms, dist.probe_sharded): gradients are reduce-scattered, each rank runs
        Adam on 1/world of every bucket and the updated weights are all-gathered (fp32 masters of the encoders, the 16-bit
        copy alone for the `inter_` heads) -- checkpoint() / save_checkpoint() / resume() are then COLLECTIVE calls.
        broadcast_from_rank0: with more than one rank the constructor is COLLECTIVE like DistributedDataParallel's
        (ssl_train.py:170): parameters, BatchNorm buffers and the GradScaler state of rank 0 overwrite every other rank's."""
        _lib.load()
        dev = next(model.parameters()).device
        if dev.type != "cuda":
            raise _lib.MsfwsiHipError("PretrainStep needs the model on a HIP device (model.cuda()); no CPU path")
        if dtype not in (torch.float32, torch.bfloat16, torch.float16):
            raise _lib.MsfwsiHipError(f"unsupported compute dtype {dtype}")
        if loss not in ("cosine", "infonce"):
            raise ValueError("loss must be 'cosine' (the reference, tools/ssl_train.py:448-466) or 'infonce'")
        # "infonce": the variant BASELINE.json's north_star names -- rows of p against the z of ALL ranks (all-gather),
        # positives on the diagonal, cross entropy at `temperature`.  The reference has no such code, so
        # this mode is parity-UNPINNED (checked against a torch restatement); the default stays the reference's loss.
        self.loss_kind, self.temperature = loss, float(temperature)
        self.model = model
        self.arch = arch
        self.dtype = dtype
        self.device = dev
        self.group = process_group
        self.weights_per_scale = tuple(float(w) for w in fuser_weights)
        self.init_lr = lr * math.sqrt(global_batch) / math.sqrt(32)  # ssl_train.py:155 (GLOBAL batch)
        self.lrs = [self.init_lr * float(m) for m in ms_lr]
        self.betas = (0.9, 0.999)
        self.eps = [1e-8, 1e-8, 1e-8]
        self.flats = FlatGroups(model, lowp_dtype=None if dtype == torch.float32 else dtype)
        multi = world_size(process_group) > 1 or os.environ.get("MSFWSI_FORCE_SYNC", "0") != "0"
        if multi and broadcast_from_rank0:
            # the DDP constructor's broadcast (ssl_train.py:170).  The reference seeds the parent process only (:46-48): its
            # mp.spawn workers (:68) build their heads from their own RNG state and THIS exchange makes the replicas equal.
            # The parameters are views of the flat buffers, so three large messages carry them; before the 16-bit copies are
            # cast (_register_lowp_weights below)
            broadcast_state(list(self.flats.w) + list(model.buffers()), process_group)
        self.engine = Engine(process_group=process_group, sync_bn=sync_bn)
        self.engine.allow_multistream = True  # flat, pre-allocated gradient accumulators: several streams may add into them
        model._engine = self.engine
        self._register_lowp_weights()
        self.grads = _FlatGradStore(self.flats)
        # gradients travel on their OWN communicator: the multi-GB all-reduce of the head group must not sit in
        # front of the latency-bound SyncBN exchanges of the encoder backward that is still running
        grad_group = process_group
        force = os.environ.get("MSFWSI_FORCE_SYNC", "0") != "0"
        if process_group is None and (world_size(process_group) > 1 or force):
            import torch.distributed as dist

            if not (dist.is_available() and dist.is_initialized()):
                raise _lib.MsfwsiHipError("MSFWSI_FORCE_SYNC needs an initialised process group")

            grad_group = dist.new_group(backend=dist.get_backend())
        if shard_optimizer is None:
            shard_optimizer = os.environ.get("MSFWSI_SHARD_OPT", "1") != "0"
        if shard_optimizer and multi:
            # the sharded exchange uses RCCL's IN-PLACE reduce-scatter / all-gather forms: checked once, on small buffers
            # with known values, with a COLLECTIVE verdict -- a backend that rejects or miscomputes them falls back to the
            # all-reduce form on every rank alike instead of failing (or diverging) inside the first step
            shard_optimizer = probe_sharded(grad_group, dev)
        self.shard_optimizer = bool(shard_optimizer) and multi
        self.reducer = GradReducer(self.flats, grad_group, shard=bool(shard_optimizer))
        if dtype != torch.float32:
            self.lazy_master_groups = (2,)  # the `inter_` heads: kernels read their 16-bit copy only
        self._init_optimizer_state(use_scaler, init_scale)
        if multi:
            if broadcast_from_rank0:
                broadcast_state([self.scale, self.growth_tracker, self.step_dev], process_group)
            # the side streams' communicators are created HERE, at a point every rank passes in the same order, not lazily
            # inside a step's forward where a rank that plans differently could desynchronise new_group
            self.engine.prepare_multirank(dev)

        def _state_dict_guard(module, prefix, keep_vars):
            if self._master_stale:
                raise RuntimeError("state_dict() after sharded optimizer steps: the fp32 master weights of the `inter_` heads "
                                   "are current only on their owning ranks -- call PretrainStep.sync_master_weights() (or "
                                   "checkpoint()) on EVERY rank first")
        model.register_state_dict_pre_hook(_state_dict_guard)
        # epoch meter of ssl_train.py:421,467-468,483-486: [sum loss*bs, sum bs], kept on the device
        self.epoch_meter = torch.zeros(2, dtype=torch.float64, device=dev)

    # ---------------------------------------------------------------------------------------
    def forward_loss(self, batch, want_grad: bool = True):
        """forward + loss (+ dLoss/dp).  batch = ((ctx_v1, ctx_v2), (tgt_v1, tgt_v2), [idx_v1, idx_v2]) with
        tgt_v* already flattened to [B*K,3,H,W] (ssl_train.py:433-438)."""
        (c1, c2), (t1, t2), idx = batch
        outs, rec = self.engine.model_forward(self.model, (c1, t1), (c2, t2), idx, self.dtype)
        self.loss_accum.zero_()
        dps: Dict[Tuple[str, int, int], torch.Tensor] = {}
        ls = self.scale if self.use_scaler else None
        for gi, grp in enumerate(("context", "target", "inter")):
            p1s, p2s, z1s, z2s = outs[gi]
            for s in range(4):
                rows = p1s[s].shape[0]
                coef = -0.5 * self.weights_per_scale[s] / rows
                for v, (p, z) in enumerate(((p1s[s], z2s[s]), (p2s[s], z1s[s]))):
                    if self.loss_kind == "infonce":
                        dp = self._infonce_term(p, z, 0.5 * self.weights_per_scale[s] / rows, ls, want_grad)
                    else:
                        dp = torch.empty_like(p) if want_grad else None
                        kn.cosine_loss(p, z, coef, self.loss_accum, dp, ls)
                    if want_grad:
                        dps[(grp, s, v)] = dp
        return outs, rec, dps

    def _infonce_term(self, p, z, coef: float, ls, want_grad: bool):
        """coef * sum_rows CE(normalize(p) . normalize(z_all)^T / tau, own row) -> loss_accum; returns dLoss/dp.
        z_all = this rank's z rows preceded / followed by the other ranks' (RCCL all-gather): the cross-GPU negative
        set; z carries no gradient (stop-gradient, backbone.py:188-191)."""
        import torch.distributed as dist

        rows, dd = p.shape
        dev = p.device
        world = world_size(self.group)
        z_all, label0 = z, 0
        if world > 1:
            z_all = torch.empty(world * rows, dd, dtype=z.dtype, device=dev)
            dist.all_gather_into_tensor(z_all, z.contiguous(), group=self.group)
            label0 = dist.get_rank(self.group) * rows
        n = z_all.shape[0]
        if n % 4 or dd % 4:
            raise ValueError(f"InfoNCE: the negative set ({n} rows = world x per-rank rows) and the embedding width ({dd}) "
                             f"must be multiples of 4 (16-byte fp32 chunks of the logit matrix)")
        ph, zh = torch.empty_like(p), torch.empty_like(z_all)
        pinv = torch.empty(rows, dtype=torch.float32, device=dev)
        zinv = torch.empty(n, dtype=torch.float32, device=dev)
        kn.row_l2norm(p, ph, pinv)
        kn.row_l2norm(z_all, zh, zinv)
        # the logit matrix and its gradient stay in fp32 whatever the storage type: a cosine in [-1, 1] rounded to bf16
        # (4e-3) times 1/tau = 5 would put 2e-2 of error on every logit before the softmax
        ph32 = ph if self.dtype == torch.float32 else kn.upcast_f32(ph)
        zh32 = zh if self.dtype == torch.float32 else kn.upcast_f32(zh)
        d = kn.conv_desc(torch.float32, rows, 1, 1, dd, n, 1, 1, 1, 0)
        logits = torch.empty(rows, n, dtype=torch.float32, device=dev)
        kn.conv_fwd(d, ph32, zh32, logits)                  # logits[i][j] = <p_hat_i, z_hat_j>
        kn.softmax_ce(logits, label0, 1.0 / self.temperature, coef, self.loss_accum, ls, write_grad=want_grad)
        if not want_grad:
            return None
        dph32 = torch.empty(rows, dd, dtype=torch.float32, device=dev)
        kn.conv_dgrad(d, logits, zh32, dph32)               # d p_hat = dlogits . z_hat
        dph = dph32 if self.dtype == torch.float32 else kn.cast_lowp(dph32, torch.empty_like(p))
        dp = torch.empty_like(p)
        kn.row_l2norm_bwd(ph, dph, pinv, dp)
        return dp

    def step(self, batch) -> torch.Tensor:
        """one optimisation step; returns the (device-resident, fp64) loss of this minibatch"""
        bs = batch[0][0].shape[0]
        self.engine.reset_counters()
        self.flats.zero_grads()
        kn.ARENA.begin_step(self.device)  # one clear for all of this step's small zero-initialised accumulators
        try:
            outs, rec, dps = self.forward_loss(batch, want_grad=True)
            loss = self.loss_accum.clone()
            self.engine.model_backward(self.model, rec, dps, self.grads, self.dtype,
                                       on_group_done=self.reducer.launch)
        finally:
            kn.ARENA.end_step()
        self.reducer.wait()
        self.engine.close_counters()
        self.optimizer_step()
        self.epoch_meter[0] += loss[0] * bs
        self.epoch_meter[1] += bs
        return loss

    def epoch_loss(self) -> float:
        """sample-weighted mean loss over ranks (ssl_train.py:483-486); syncs once per epoch"""
        m = self.epoch_meter.clone()
        if world_size(self.group) > 1:
            import torch.distributed as dist

            dist.all_reduce(m, group=self.group)
        self.epoch_meter.zero_()
        return float(m[0] / m[1])

    def checkpoint(self, epoch: int) -> dict:
        """the dict the reference passes to save_checkpoint (ssl_train.py:375-386); DDP's "module." prefix kept"""
        self.sync_master_weights()  # (sharded optimizer: a collective -- every rank calls checkpoint(), rank 0 saves)
        sd = {"module." + k: v.detach().clone() for k, v in self.model.state_dict().items()}
        return {"epoch": epoch + 1, "arch": self.arch, "state_dict": sd, "optimizer": self.optimizer_state_dict(),
                "scaler": self.scaler_state_dict()}

    def save_checkpoint(self, path: str, epoch: int):
        """ssl_train.py:375-386,489-492.  With more than one rank this is a COLLECTIVE call: every rank calls it -- OUTSIDE
        the reference's `if rank == 0` guard (:363), because the sharded optimizer's master weights and Adam moments are
        gathered in checkpoint() -- rank 0 alone writes the file, and all ranks leave together (a barrier), so no rank
        reads or resumes from a half-written file."""
        ckpt = self.checkpoint(epoch)
        multi = world_size(self.group) > 1
        if not multi:
            torch.save(ckpt, path)
            return
        import torch.distributed as dist

        if dist.get_rank(self.group) == 0:
            torch.save(ckpt, path)
        del ckpt
        dist.barrier(group=self.group)

    def resume(self, ckpt: dict) -> int:
        """ssl_train.py:313-335 incl. the hard-coded eps=0.1 after loading"""
        self._master_stale = False  # every master is about to be overwritten
        sd = {(k[len("module."):] if k.startswith("module.") else k): v for k, v in ckpt["state_dict"].items()}
        self.model.load_state_dict(sd)  # copies into the flat-buffer views in place
        self.load_optimizer_state_dict(ckpt["optimizer"])
        self.eps = [0.1, 0.1, 0.1]
        self.load_scaler_state_dict(ckpt.get("scaler", {}))
        if self.dtype != torch.float32:
            for gi in range(len(self.flats.w)):
                kn.cast_lowp(self.flats.w[gi], self.flats.w16[gi])
        self.engine.invalidate_weights()
        return int(ckpt["epoch"])


def synthetic_batch(B: int, size: int = 224, K: int = 16, seed: int = 0, device="cuda"):
    """Synthetic input with the reference's batch contract (src/utils/data/bcss.py:164-182 after collate and
    the H2D of ssl_train.py:430-438): N(0,1) images and inverse jigsaw permutations, generated on the device."""
    g = torch.Generator(device=device).manual_seed(seed)
    mk = lambda n: torch.randn(n, 3, size, size, generator=g, device=device)
    c1, c2, t1, t2 = mk(B), mk(B), mk(B * K), mk(B * K)
    cg = torch.Generator().manual_seed(seed)
    idx = [torch.stack([torch.argsort(torch.randperm(K, generator=cg)) for _ in range(B)]) for _ in range(2)]
    return (c1, c2), (t1, t2), idx
