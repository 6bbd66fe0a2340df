"""Data-parallel host logic for the MSF-WSI pre-train step: one process per GPU, torch.distributed
("nccl" == RCCL over xGMI on ROCm; "gloo" on CPU for tests).

What the reference does (tools/ssl_train.py:160-170, :262-275): DistributedSampler shards whole samples,
SyncBatchNorm makes an N-rank step identical to a 1-rank step on the concatenated batch, DDP averages
gradients with 25 MB buckets.  Here:
  * `FlatGroups`   lays the three optimizer groups (context_/target_/inter_, ssl_train.py:281-300) out as flat
                   fp32 buffers (weights, grads, Adam moments [+ bf16 copy]) so one kernel / one collective
                   covers a whole group; parameters become views (conv weights keep channels_last order).
  * `GradReducer`  launches one asynchronous all-reduce per group as soon as the backward schedule finishes
                   it (heads first = 80 % of the bytes), i.e. large messages that RCCL can spread over all
                   seven xGMI links, overlapped with the remaining encoder backward.
  * `sync_sums`    the cross-replica BatchNorm exchange: a SUM all-reduce of packed fp64 [sum, sumsq] vectors
                   (sums are associative, so the result equals full-batch statistics exactly).
None of this touches the GPU directly: it is plain tensor/collective plumbing and runs under gloo on CPU.
"""
from __future__ import annotations

import os
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist
import torch.nn as nn

GROUP_PREFIXES = ("context_", "target_", "inter_")
ALIGN = 64  # elements: every parameter starts on a 256-byte boundary of the flat buffer


def world_size(group=None) -> int:
    return dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1


def rank(group=None) -> int:
    return dist.get_rank(group) if dist.is_available() and dist.is_initialized() else 0


def shard_range(n_samples: int, world: int, r: int) -> Tuple[int, int]:
    """contiguous per-rank slice of whole samples (a sample = 1 context + K target tiles, both views)"""
    if n_samples % world != 0:
        raise ValueError(f"global batch {n_samples} is not divisible by world size {world} (drop_last semantics)")
    per = n_samples // world
    return r * per, (r + 1) * per


def sync_sums(packed: torch.Tensor, group=None, force: bool = False) -> torch.Tensor:
    """The cross-replica BatchNorm exchange: in-place SUM all-reduce of a packed fp64 statistics vector ([sum, sumsq]
    forward, [sum g, sum g*c] backward).  The engine (GPU tensors, RCCL) and tests/test_dist_cpu.py (CPU tensors, gloo)
    both go through this function.  force: issue the collective even with one rank (RCCL rehearsal)."""
    if world_size(group) > 1 or (force and dist.is_available() and dist.is_initialized()):
        dist.all_reduce(packed, op=dist.ReduceOp.SUM, group=group)
    return packed


_PROBED = set()


def probe_collectives(group=None, device=None):
    """One-time check that the process group's backend executes every collective form this package issues:
    fp64 SUM (packed BatchNorm statistics), fp32 AVG -- or SUM where AVG does not exist (gloo) -- on a flat gradient
    buffer, int32 MAX (the collective recompute plan).  Raises RuntimeError naming the missing capability instead of
    failing somewhere inside a training step.  Returns the reduce op to use for gradient averaging."""
    backend = dist.get_backend(group)
    key = (id(group), backend)
    dev = device if device is not None else ("cuda" if backend == "nccl" else "cpu")
    avg = backend == "nccl"
    if key in _PROBED:
        return dist.ReduceOp.AVG if avg else dist.ReduceOp.SUM
    w = dist.get_world_size(group)
    try:
        a = torch.ones(8, dtype=torch.float64, device=dev)
        dist.all_reduce(a, op=dist.ReduceOp.SUM, group=group)
        b = torch.ones(8, dtype=torch.float32, device=dev)
        dist.all_reduce(b, op=dist.ReduceOp.AVG if avg else dist.ReduceOp.SUM, group=group)
        c = torch.full((1,), dist.get_rank(group), dtype=torch.int32, device=dev)
        dist.all_reduce(c, op=dist.ReduceOp.MAX, group=group)
        ok = (float(a[0]) == float(w) and float(b[0]) == (1.0 if avg else float(w)) and int(c[0]) == w - 1)
    except Exception as e:  # noqa: BLE001 -- any backend error is the finding
        raise RuntimeError(f"process-group backend '{backend}' cannot run a collective the MSF-WSI data-parallel path "
                           f"needs (fp64 SUM / fp32 {'AVG' if avg else 'SUM'} / int32 MAX all-reduce): {e}") from e
    if not ok:
        raise RuntimeError(f"process-group backend '{backend}' returned wrong results in the collective probe "
                           f"(fp64 SUM {float(a[0])}, fp32 {float(b[0])}, int32 MAX {int(c[0])}; world {w})")
    _PROBED.add(key)
    return dist.ReduceOp.AVG if avg else dist.ReduceOp.SUM


def broadcast_state(tensors: Sequence[torch.Tensor], group=None, src: int = 0, coalesce_below: int = 1 << 22):
    """The DDP constructor's exchange (tools/ssl_train.py:170: `_sync_module_states` -- parameters AND buffers of rank 0
    overwrite every other rank's): in place.  It is load-bearing in the reference: main() seeds the parent process only
    (:46-48), the mp.spawn workers (:68) build their model from their own RNG state, and this broadcast is what makes the
    replicas equal.  `src` is a rank of `group`.  Large tensors (the flat weight buffers of the optimizer groups) travel
    as they are, small ones (BatchNorm buffers, scaler state) coalesced per dtype into one message each."""
    if not (dist.is_available() and dist.is_initialized()):
        return
    gsrc = dist.get_global_rank(group, src) if group is not None else src
    small: Dict[Tuple[torch.dtype, torch.device], List[torch.Tensor]] = {}
    for t in tensors:
        if t.numel() == 0:
            continue
        if t.numel() >= coalesce_below and t.is_contiguous():
            dist.broadcast(t, src=gsrc, group=group)
        else:
            small.setdefault((t.dtype, t.device), []).append(t)
    for (dtype, dev), ts in small.items():
        flat = torch.cat([t.detach().reshape(-1) for t in ts])
        dist.broadcast(flat, src=gsrc, group=group)
        pos = 0
        with torch.no_grad():
            for t in ts:
                t.copy_(flat[pos:pos + t.numel()].view(t.shape))
                pos += t.numel()


def probe_sharded(group=None, device=None) -> bool:
    """One-time check of the sharded optimizer's collective forms on this backend: the IN-PLACE reduce_scatter_tensor (output =
    this rank's slice of the input) and the in-place all_gather_into_tensor (input = this rank's slice of the output), on a
    small buffer with known values.  The verdict is COLLECTIVE (MIN over ranks of the local result; a backend that rejects
    the form raises on every rank alike), so every rank takes the same branch: True -> reduce-scatter / all-gather,
    False -> the all-reduce form (GradReducer(shard=False))."""
    backend = dist.get_backend(group)
    dev = device if device is not None else ("cuda" if backend == "nccl" else "cpu")
    w, r = dist.get_world_size(group), dist.get_rank(group)
    ok = 1
    try:
        per = 8
        buf = torch.arange(per * w, dtype=torch.float32, device=dev) + 1.0
        dist.reduce_scatter_tensor(buf[r * per:(r + 1) * per], buf, op=dist.ReduceOp.SUM, group=group)
        want = (torch.arange(r * per, (r + 1) * per, dtype=torch.float32, device=dev) + 1.0) * w
        ok &= int(torch.equal(buf[r * per:(r + 1) * per], want))
        out = torch.zeros(per * w, dtype=torch.bfloat16, device=dev)
        out[r * per:(r + 1) * per] = float(r + 1)
        dist.all_gather_into_tensor(out, out[r * per:(r + 1) * per], group=group)
        want = torch.arange(1, w + 1, dtype=torch.bfloat16, device=dev).repeat_interleave(per)
        ok &= int(torch.equal(out, want))
    except Exception:  # noqa: BLE001 -- an unsupported form is the finding
        ok = 0
    flag = torch.full((1,), ok, dtype=torch.int32, device=dev)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
    return bool(int(flag[0]))


class FlatGroups:
    """Flat storage of the model parameters by optimizer group."""

    def __init__(self, model: nn.Module, with_bf16: bool = False, device=None, lowp_dtype=None,
                 prefixes: Sequence[str] = GROUP_PREFIXES):
        """prefixes: name prefixes of the optimizer groups, in order (the pre-train loop's three, ssl_train.py:281-300;
        ("",) = one group with every parameter, as the fine-tune loop's optim.Adam(model.parameters()),
        ssl_finetune.py:289)"""
        if with_bf16 and lowp_dtype is None:
            lowp_dtype = torch.bfloat16
        with_bf16 = lowp_dtype is not None
        named = list(model.named_parameters())
        self.prefixes = tuple(prefixes)
        self.names: List[List[str]] = [[n for n, _ in named if n.startswith(p)] for p in self.prefixes]
        self.params: List[List[nn.Parameter]] = [[p for n, p in named if n.startswith(pre)] for pre in self.prefixes]
        covered = sum(len(g) for g in self.params)
        if covered != len(named):
            raise ValueError("every parameter must belong to exactly one of the optimizer groups " + repr(self.prefixes))
        self.offsets: List[List[int]] = []
        self.sizes: List[int] = []
        self.w: List[torch.Tensor] = []
        self.g: List[torch.Tensor] = []
        self.m: List[torch.Tensor] = []
        self.v: List[torch.Tensor] = []
        self.w16: List[Optional[torch.Tensor]] = []
        for plist in self.params:
            offs, total = [], 0
            for p in plist:
                offs.append(total)
                total += (p.numel() + ALIGN - 1) // ALIGN * ALIGN
            dev = device if device is not None else plist[0].device
            self.offsets.append(offs)
            self.sizes.append(total)
            self.w.append(torch.zeros(total, dtype=torch.float32, device=dev))
            self.g.append(torch.zeros(total, dtype=torch.float32, device=dev))
            self.m.append(torch.zeros(total, dtype=torch.float32, device=dev))
            self.v.append(torch.zeros(total, dtype=torch.float32, device=dev))
            self.w16.append(torch.zeros(total, dtype=lowp_dtype, device=dev) if with_bf16 else None)
        self._grad_views: Dict[int, torch.Tensor] = {}
        self._w16_views: Dict[int, torch.Tensor] = {}
        self._stored: Dict[int, set] = {}
        for gi, plist in enumerate(self.params):
            for p, off in zip(plist, self.offsets[gi]):
                phys_shape = self.physical_shape(p)
                n = p.numel()
                src = p.data.permute(0, 2, 3, 1) if p.dim() == 4 else p.data
                self.w[gi][off:off + n].view(phys_shape).copy_(src)
                p.data = self._logical(self.w[gi][off:off + n].view(phys_shape), p)
                self._grad_views[id(p)] = self.g[gi][off:off + n].view(phys_shape)
                if with_bf16:
                    self._w16_views[id(p)] = self.w16[gi][off:off + n].view(phys_shape)

    @staticmethod
    def physical_shape(p: torch.Tensor):
        return (p.shape[0], p.shape[2], p.shape[3], p.shape[1]) if p.dim() == 4 else tuple(p.shape)

    @staticmethod
    def _logical(phys: torch.Tensor, p: torch.Tensor) -> torch.Tensor:
        return phys.permute(0, 3, 1, 2) if p.dim() == 4 else phys

    def grad_view(self, p: torch.Tensor) -> torch.Tensor:
        return self._grad_views[id(p)]

    def w16_view(self, p: torch.Tensor) -> Optional[torch.Tensor]:
        return self._w16_views.get(id(p))

    def state_views(self, gi: int, pi: int):
        """(exp_avg, exp_avg_sq) of one parameter in its logical shape"""
        p = self.params[gi][pi]
        off, n = self.offsets[gi][pi], p.numel()
        shp = self.physical_shape(p)
        return (self._logical(self.m[gi][off:off + n].view(shp), p), self._logical(self.v[gi][off:off + n].view(shp), p))

    def zero_grads(self, skip_groups: Sequence[int] = ()):
        """clear the gradient accumulators -- except the ranges `mark_stored` named: tensors whose gradient is WRITTEN by
        one launch per step (msfwsi_conv_wgrad_store: the fuser heads' 18432-wide matrices, 4 GB of fp32 together)"""
        for gi in range(len(self.g)):
            if gi not in skip_groups:
                self.zero_group(gi)

    def zero_group(self, gi: int):
        g = self.g[gi]
        skip = sorted(self._stored.get(gi, ()))
        if not skip:
            g.zero_()
            return
        pos = 0
        for lo, hi in skip + [(g.numel(), g.numel())]:
            if lo > pos:
                g[pos:lo].zero_()
            pos = max(pos, hi)

    def mark_stored(self, p: torch.Tensor):
        """the engine stores (not accumulates) this parameter's gradient, every step: zero_grads may leave it alone"""
        for gi, plist in enumerate(self.params):
            for pi, q in enumerate(plist):
                if q is p:
                    lo = self.offsets[gi][pi]
                    self._stored.setdefault(gi, set()).add((lo, lo + p.numel()))
                    return
        raise KeyError("parameter is not part of the flat groups")

    def unmark_stored(self, p: torch.Tensor) -> bool:
        """the engine ACCUMULATES into this parameter's gradient this step: if an earlier step stored it, zero_grads has
        skipped its range and it still holds that step's gradient -- clear it now and let zero_grads cover it again.
        True when the parameter had been marked."""
        for gi, plist in enumerate(self.params):
            for pi, q in enumerate(plist):
                if q is p:
                    lo = self.offsets[gi][pi]
                    rng = (lo, lo + p.numel())
                    if rng in self._stored.get(gi, ()):
                        self._stored[gi].discard(rng)
                        self.g[gi][rng[0]:rng[1]].zero_()
                        return True
                    return False
        raise KeyError("parameter is not part of the flat groups")

    def range_of(self, gi: int, name_prefix: str) -> Tuple[int, int]:
        """[lo, hi) element range of group gi's flat buffers that holds the parameters whose names start with
        `name_prefix` (e.g. "inter_projector.3."): they are adjacent in named_parameters() order, hence contiguous"""
        idx = [i for i, n in enumerate(self.names[gi]) if n.startswith(name_prefix)]
        if not idx or idx != list(range(idx[0], idx[-1] + 1)):
            raise ValueError(f"no contiguous parameter range for prefix {name_prefix!r} in group {self.prefixes[gi]!r}")
        lo = self.offsets[gi][idx[0]]
        hi = self.offsets[gi][idx[-1] + 1] if idx[-1] + 1 < len(self.offsets[gi]) else self.sizes[gi]
        return lo, hi


def shard_bucket(lo: int, hi: int, world: int, r: int, align: int = 4):
    """How one gradient bucket [lo, hi) of a flat group is split for the sharded optimizer (ZeRO-1 style): the leading
    `per * world` elements are reduce-scattered, rank r owning [lo + r*per, lo + (r+1)*per) with per = (n // world) rounded
    down to `align` elements (the Adam kernel steps in 16-byte groups); the remaining < world*(align+1) elements -- the
    TAIL -- are all-reduced and stepped by every rank (identical arithmetic everywhere).  Returns (per, own, tail) with
    own / tail element ranges (own empty when the bucket is shorter than world*align)."""
    n = hi - lo
    per = (n // world) // align * align
    main = per * world
    own = (lo + r * per, lo + (r + 1) * per)
    return per, own, (lo + main, hi)


class GradReducer:
    """Per-group asynchronous gradient averaging (the DDP reduction of tools/ssl_train.py:170).

    `launch(group)` averages a whole optimizer group in one collective; `launch(group, part=prefix)` only the slice of
    it that holds the parameters named `prefix*` -- the backward schedule calls it per scale of the `inter_` heads (80-95 %
    of all gradient bytes: 4 x (projector + predictor) = 8 buckets, the largest 4 GB), each as soon as that scale's weight
    gradients are complete, so RCCL starts on the first bucket while the next head is still in backward and the exchange
    is several medium messages instead of one 6.3 GB one (DDP buckets at 25 MB, ssl_train.py:170; xGMI rings are per-link
    bound, and nothing is gained below ~100 MB per message).  `launch(group)` after parts sends what is left of it."""

    def __init__(self, flats: FlatGroups, group=None, shard: bool = False):
        """shard: reduce-scatter every bucket instead of all-reducing it -- each rank receives the averaged gradient of
        1/world of the bucket (`owned`), runs Adam on that shard only and all-gathers the updated weights
        (train.FlatAdamScaler.optimizer_step): Adam's 28 B/parameter of HBM traffic and its arithmetic drop to 1/world per
        rank, and the fuser heads' 16-bit weight copy travels back instead of a second fp32 half of an all-reduce"""
        self.flats = flats
        self.group = group
        self.pending: List[Tuple[Optional[torch.Tensor], object]] = []
        self.world = world_size(group)
        self.shard = bool(shard)
        # per group: element ranges of the flat buffers whose averaged gradient THIS rank holds after wait() and must step
        # (its reduce-scattered shards and every bucket's all-reduced tail), and the (lo, per) of every scattered bucket
        self.owned: Dict[int, List[Tuple[int, int]]] = {}
        self.scattered: Dict[int, List[Tuple[int, int]]] = {}
        # MSFWSI_FORCE_SYNC: rehearse the collective path with a single rank (RCCL calls execute, results unchanged)
        self.active = self.world > 1 or (os.environ.get("MSFWSI_FORCE_SYNC", "0") != "0" and dist.is_available()
                                         and dist.is_initialized())
        self.op = probe_collectives(group, flats.g[0].device) if self.active else None
        self.sent: Dict[int, List[Tuple[int, int]]] = {}  # group index -> element ranges already launched this step
        self.launches = 0                                   # collectives launched since the last wait()
        self.bytes = 0
        self.launches_last_step = 0                         # ... and of the step that the last wait() closed
        self.bytes_last_step = 0

    def _send(self, buf: torch.Tensor, gi: int = -1, lo: int = 0):
        """exchange buf = flats.g[gi][lo : lo + len(buf)]"""
        if self.shard and gi >= 0:
            r = rank(self.group)
            per, own, tail = shard_bucket(lo, lo + buf.numel(), self.world, r)
            g = self.flats.g[gi]
            if per > 0:
                # in place: the output is this rank's slice of the input (NCCL / RCCL's in-place reduce-scatter form)
                src, dst = g[lo:lo + per * self.world], g[own[0]:own[1]]
                work = dist.reduce_scatter_tensor(dst, src, op=self.op, group=self.group, async_op=True)
                self.pending.append((None if self.op == dist.ReduceOp.AVG else dst, work))
                self.owned.setdefault(gi, []).append(own)
                self.scattered.setdefault(gi, []).append((lo, per))
                self.launches += 1
                self.bytes += src.numel() * src.element_size()
            if tail[1] > tail[0]:
                t = g[tail[0]:tail[1]]
                work = dist.all_reduce(t, op=self.op, group=self.group, async_op=True)
                self.pending.append((None if self.op == dist.ReduceOp.AVG else t, work))
                self.owned.setdefault(gi, []).append(tail)
                self.launches += 1
                self.bytes += t.numel() * t.element_size()
            return
        work = dist.all_reduce(buf, op=self.op, group=self.group, async_op=True)
        self.pending.append((None if self.op == dist.ReduceOp.AVG else buf, work))
        self.launches += 1
        self.bytes += buf.numel() * buf.element_size()

    def launch(self, group_name, part: Optional[str] = None):
        if not self.active:
            return
        gi = group_name if isinstance(group_name, int) else {"context": 0, "target": 1, "inter": 2}[group_name]
        buf = self.flats.g[gi]
        if part is not None:
            lo, hi = self.flats.range_of(gi, part)
            self.sent.setdefault(gi, []).append((lo, hi))
            self._send(buf[lo:hi], gi, lo)
            return
        done = sorted(self.sent.pop(gi, []))
        pos = 0
        for lo, hi in done + [(buf.numel(), buf.numel())]:  # whatever no part covered
            if lo > pos:
                self._send(buf[pos:lo], gi, pos)
            pos = max(pos, hi)

    def wait(self):
        for buf, work in self.pending:
            work.wait()
            if buf is not None:  # backends without AVG (gloo): SUM, then scale
                buf.mul_(1.0 / self.world)
        self.pending = []
        self.sent = {}
        self.launches_last_step, self.bytes_last_step = self.launches, self.bytes
        self.launches = self.bytes = 0

    @property
    def sharding(self) -> bool:
        """this step's gradients arrived reduce-scattered: optimizer_step steps `owned` and gathers the weights"""
        return self.shard and self.active

    def take_shards(self):
        """(owned, scattered) of the step just waited for; the reducer starts the next step empty"""
        o, s = self.owned, self.scattered
        self.owned, self.scattered = {}, {}
        return o, s

    def gather_weights(self, gi: int, scattered: List[Tuple[int, int]], buf: torch.Tensor):
        """all-gather the freshly stepped shards of `buf` (a flat weight buffer of group gi, fp32 or its 16-bit copy) so
        that every rank holds the whole bucket again; in place (each rank's input is its slice of the output)"""
        r = rank(self.group)
        works = []
        for lo, per in scattered:
            out, inp = buf[lo:lo + per * self.world], buf[lo + r * per:lo + (r + 1) * per]
            works.append(dist.all_gather_into_tensor(out, inp, group=self.group, async_op=True))
            self.gather_bytes += out.numel() * out.element_size()
        return works

    gather_bytes = 0
