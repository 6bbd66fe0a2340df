"""Execution engine of the MSF-WSI pre-train step on MI355X: schedules the hand-written HIP kernels
(msf_wsi_amd/csrc via msf_wsi_amd.kernels) for the forward and the hand-derived backward of
  * the ResNet encoders                      (reference src/models/resnet.py:232-256)
  * MSFWSI.forward: 4 encoder passes, jigsaw un-shuffle, 24 MLP heads, fuser concat
                                              (reference src/models/backbone.py:129-222)
and exposes them to torch autograd as ONE node, so `loss.backward()` of the reference loop
(tools/ssl_train.py:472) lands in the same kernels and parameter `.grad`s appear as ordinary tensors
(DDP / optimizers read them unchanged).

Memory model: per conv only the RAW output c is kept (storage dtype); normalised operands relu(bn(c)) are transient
(one streaming pass where a dense kernel needs them).  Residual-block outputs are the only normalised activations
that are kept.  The 4x-wide conv3 output of a Bottleneck (and the downsample branch's output) never exist: their
BatchNorm statistics come from Gram matrices of the operands, the conv runs once with BatchNorm apply + residual +
ReLU in its epilogue, and the BatchNorm backward is folded into [K][C] weight-sized matrices.

There is no CPU / eager-torch fallback: CPU tensors raise.
"""
from __future__ import annotations

import contextlib
import os
import threading
import types
import weakref
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist
import torch.nn as nn

from . import _lib
from . import kernels as kn
from .dist import probe_collectives, sync_sums


# ------------------------------------------------------------------------------------------------
# records
# ------------------------------------------------------------------------------------------------
@dataclass
class BNState:
    scale: torch.Tensor
    shift: torch.Tensor
    mean: torch.Tensor
    invstd: torch.Tensor
    count: float  # elements per channel over ALL replicas
    frozen: bool = False  # eval-mode BatchNorm: the map comes from the running statistics (no backward through them)


@dataclass
class Unit:
    """one conv/linear (+ optional BatchNorm) application"""
    op: nn.Module
    bn: Optional[nn.Module]
    relu: bool
    desc: object
    x: torch.Tensor                  # operand tensor (raw)
    x_pro: Optional[BNState]         # BatchNorm+ReLU to apply to x on load
    c: torch.Tensor                  # raw output (None: never formed / dropped, see _conv_bn_res_fwd)
    st: Optional[BNState] = None
    gram: Optional[Tuple[torch.Tensor, torch.Tensor]] = None  # (a^T a, sum a) of the normalised operand, fp32/fp64
    lo: Optional[Tuple[int, tuple]] = None  # (stride, full-resolution shape): x is a strided subsampling of the input
    s2d: bool = False  # stem in space-to-depth form: desc / x describe the 4x4 / stride-1 conv on the [N,H/2,W/2,16] operand


@dataclass
class BlockRec:
    y_in: torch.Tensor
    units: List[Unit]
    ds: Optional[Unit]
    y_out: torch.Tensor
    HW: int
    stage: int
    stage_end: bool
    gate_bits: Optional[torch.Tensor] = None  # (y_out > 0) as one byte per 16-byte chunk (fused tail forward)


@dataclass
class EncPass:
    enc: nn.Module
    N: int
    H: int
    W: int
    xin: torch.Tensor
    stem: Unit
    pooled: torch.Tensor
    amax: torch.Tensor
    blocks: List[BlockRec]
    feats: List[torch.Tensor]
    x_src: Optional[torch.Tensor] = None  # original NCHW input, kept when the pass must be recomputed
    saved: bool = True


@dataclass
class ChainRec:
    units: List[Unit]
    out: torch.Tensor


@dataclass
class StepRec:
    B: int
    enc: Dict[str, EncPass] = field(default_factory=dict)
    idx: List[torch.Tensor] = field(default_factory=list)
    heads: Dict[Tuple[str, int, int], Tuple[ChainRec, ChainRec]] = field(default_factory=dict)
    tgt_sorted: Dict[Tuple[int, int], torch.Tensor] = field(default_factory=dict)
    nosave: set = field(default_factory=set)
    dual: bool = False  # view-1 passes ran on the side stream (their activations live in that stream's pool)
    head_streams: bool = False  # the context / target head groups ran on the context / view-1 streams (same for their records)
    tri: bool = False   # ... and both context passes on the third stream
    pair_bwd: bool = True  # the memory plan allows the two views' backward passes side by side (lockstep)


def chan_pad(dtype: torch.dtype) -> int:
    """stem input channels are zero-padded to one 16-byte chunk"""
    return 4 if dtype == torch.float32 else 8


# ------------------------------------------------------------------------------------------------
# weights / gradients
# ------------------------------------------------------------------------------------------------
class WeightStore:
    """Compute-dtype operand views of the parameters.  fp32 uses the parameter storage in place
    (channels_last == [K][R][S][C]); bf16 copies are cast by the HIP cast kernel and cached per parameter
    version.  A trainer may pre-register externally maintained copies (`register`).

    Every entry is keyed on the `id` of the tensor it derives from AND carries a weak reference to it: an entry whose
    source object has died never answers for a new object that happens to get the same id, version counter and device
    address (the process-wide default engine outlives the models it serves -- without the check a second model built
    after the first was dropped could be multiplied with the first one's weights)."""

    def __init__(self):
        self._cache: Dict[tuple, tuple] = {}    # (id(param), dtype, pad) -> (weakref(param), version key, copy)
        self._ext: Dict[tuple, tuple] = {}      # (id(param), dtype)      -> (weakref(param), copy)
        self._derived: Dict[tuple, tuple] = {}  # (kind, id(source))      -> (weakref(source), tensor)

    def register(self, param: torch.Tensor, dtype: torch.dtype, tensor: torch.Tensor):
        self._ext[(id(param), dtype)] = (weakref.ref(param), tensor)

    def clear(self):
        """drop every derived copy (not the registered ones: their owner keeps them current)"""
        self._cache.clear()
        self._derived.clear()

    @staticmethod
    def physical(param: torch.Tensor) -> torch.Tensor:
        """fp32 [K][R][S][C] (or [out][in]) contiguous view of a parameter; re-lays it out once if needed"""
        if param.dim() == 4:
            v = param.data.permute(0, 2, 3, 1)
            if not v.is_contiguous():
                param.data = param.data.contiguous(memory_format=torch.channels_last)
                v = param.data.permute(0, 2, 3, 1)
                if not v.is_contiguous():  # degenerate strides (size-1 dims): force a real copy
                    param.data = param.data.permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)
                    v = param.data.permute(0, 2, 3, 1)
            return v
        return param.data if param.data.is_contiguous() else param.data.contiguous()

    def get(self, param: torch.Tensor, dtype: torch.dtype, pad_to: int = 0) -> torch.Tensor:
        ext = self._ext.get((id(param), dtype))
        if ext is not None and ext[0]() is param:
            return ext[1]
        phys = self.physical(param)
        if dtype == torch.float32 and not pad_to:
            return phys
        key = (id(param), dtype, pad_to)
        ver = (param._version, phys.data_ptr())
        hit = self._cache.get(key)
        if hit is not None and hit[0]() is param and hit[1] == ver:
            return hit[2]
        if pad_to:
            rows = phys.numel() // phys.shape[-1]
            out = torch.empty(*phys.shape[:-1], pad_to, dtype=dtype, device=phys.device)
            kn.pad_cast(phys, out, rows, phys.shape[-1], pad_to)
        else:
            out = torch.empty(phys.shape, dtype=dtype, device=phys.device)
            kn.cast_lowp(phys, out)
        self._cache[key] = (weakref.ref(param), ver, out)
        if len(self._cache) > 4096:  # entries of dead parameters (models dropped while the engine lives on)
            self._cache = {k: v for k, v in self._cache.items() if v[0]() is not None}
        return out

    def derived(self, kind: str, src: torch.Tensor, make) -> torch.Tensor:
        """a tensor computed from `src` (a cached or registered copy), kept while that very object lives and until
        `clear`: owners that rewrite `src` through raw pointers (the flat Adam kernel) call `clear` after each update"""
        key = (kind, id(src))
        hit = self._derived.get(key)
        if hit is not None and hit[0]() is src:
            return hit[1]
        out = make(src)
        self._derived[key] = (weakref.ref(src), out)
        if len(self._derived) > 4096:
            self._derived = {k: v for k, v in self._derived.items() if v[0]() is not None}
        return out


class GradStore:
    """fp32 gradient accumulators in the kernels' physical layout, created zeroed on first use."""

    def __init__(self):
        self.bufs: Dict[int, torch.Tensor] = {}
        self.params: Dict[int, torch.Tensor] = {}
        self._lock = threading.Lock()  # two view passes may run their backward on two host threads (_ViewPair)

    def get(self, param: torch.Tensor) -> torch.Tensor:
        b = self.bufs.get(id(param))
        if b is None:
            with self._lock:
                b = self.bufs.get(id(param))
                if b is None:
                    phys = WeightStore.physical(param)
                    b = torch.zeros(phys.shape, dtype=torch.float32, device=phys.device)
                    self.bufs[id(param)] = b
                    self.params[id(param)] = param
        return b

    def logical(self, param: torch.Tensor) -> Optional[torch.Tensor]:
        b = self.bufs.get(id(param))
        if b is None:
            return None
        return b.permute(0, 3, 1, 2) if param.dim() == 4 else b


class _ViewPair:
    """Rendezvous of the two view passes of ONE encoder running in lockstep on two host threads (Engine._run_views).

    Why: with SyncBatchNorm every BatchNorm call exchanges one small packed fp64 message per direction, and the two
    views of an encoder are separate BatchNorm batches (backbone.py:140-145) that issue the SAME sequence of exchanges.
    Run one after the other that is 4 x 154 = 616 latency-bound all-reduces per ResNet-50 step; run in lockstep, both
    views' vectors travel in ONE message per BatchNorm and direction (308).  The passes still enqueue on one HIP stream:
    a thread here is a coroutine with a stack, not a second stream -- kernels of the two views interleave in launch
    order, data dependencies inside a view follow that view's program order, and the collective of an exchange is
    enqueued by whichever view arrives second, i.e. after both producers and before either consumer.

    exchange(v, fn): both views call it at the same point of their sequence; the last arriver runs fn() (the one
    collective); everybody leaves after it has been enqueued.  turn(v): view 1 waits until view 0 has passed the same
    point (the running-statistics update of a module happens for view 0 first, as in the reference)."""

    def __init__(self):
        self.cv = threading.Condition()
        self.count = [0, 0]      # exchanges entered per view
        self.done = 0            # exchanges whose collective has been enqueued
        self.turns = 0           # turn points view 0 has passed
        self.tcount = [0, 0]
        self.failed: Optional[BaseException] = None

    def fail(self, e: BaseException):
        with self.cv:
            if self.failed is None:
                self.failed = e
            self.cv.notify_all()

    def _check(self):
        if self.failed is not None:
            raise RuntimeError("the other view pass of this encoder failed") from self.failed

    def exchange(self, v: int, fn):
        with self.cv:
            self._check()
            self.count[v] += 1
            seq = self.count[v]
            if self.count[1 - v] >= seq:  # the other view is already here: this thread issues the collective
                try:
                    fn()
                except BaseException as e:  # noqa: BLE001 -- wake the partner, then re-raise here
                    self.failed = e
                    self.cv.notify_all()
                    raise
                self.done = seq
                self.cv.notify_all()
                return
            while self.done < seq:
                if not self.cv.wait(timeout=600.0):
                    self.failed = TimeoutError("view passes out of step: the partner never reached exchange %d" % seq)
                self._check()

    def turn_wait(self, v: int):
        """view 1: block until view 0 has passed its matching turn point; view 0: nothing to wait for"""
        with self.cv:
            self.tcount[v] += 1
            if v == 0:
                return
            while self.turns < self.tcount[1]:
                if not self.cv.wait(timeout=600.0):
                    self.failed = TimeoutError("view passes out of step at a turn point")
                self._check()

    def turn_done(self, v: int):
        if v == 0:
            with self.cv:
                self.turns += 1
                self.cv.notify_all()


# ------------------------------------------------------------------------------------------------
class Engine:
    # Dispatch / schedule options: plain attributes with the measured-best defaults (each A/B is recorded in DESIGN.md and
    # profiles/).  They are NOT environment switches any more (round 5 had 57 MSFWSI_* variables); code and tests set the
    # attribute, and A/B scripts pass ONE variable, MSFWSI_ENGINE="name=value,name=value", checked against this table.
    # tests/test_options_gpu.py flips every one of them on a whole training step.
    OPTIONS = ("halo3x3", "fuse_pro3x3", "fold_bn3", "fuse_gate", "fold_bn3_fwd", "gate_bits", "fuse_two_source",
               "dgrad2_pro", "dgrad2_pro_max_c", "panel_fwd", "panel_dgrad", "heads_on_streams", "img3x3", "img3x3_layer1", "img3x3_s2", "gap_stride_fused",
               "fuse_a2_wgrad", "fuse_a2_wgrad_max_c", "img3x3_chunk_bytes", "img3x3_min_fill", "panel_gram", "panel_fwd_min_k", "stem_run",
               "stem_s2d", "stem_fuse_bnbwd", "pair_head_wgrad", "pair_head_fwd", "bucket_inter", "store_head_wgrad", "fold_ds",
               "fold_ds_fwd", "fold_ds_strided", "lores_resid", "ctx_stream", "coalesce_views")

    def _apply_env_options(self):
        spec = os.environ.get("MSFWSI_ENGINE", "").strip()
        if not spec:
            return
        for item in spec.split(","):
            name, sep, val = item.strip().partition("=")
            if not sep or name not in self.OPTIONS:
                raise _lib.MsfwsiHipError(f"MSFWSI_ENGINE: unknown option {item!r} (options: {', '.join(self.OPTIONS)})")
            cur = getattr(self, name)
            setattr(self, name, (val not in ("0", "false", "False")) if isinstance(cur, bool) else type(cur)(float(val)))

    def __init__(self, process_group=None, sync_bn: Optional[bool] = None):
        self.weights = WeightStore()
        self.group = process_group
        self._sync_bn = sync_bn
        self.update_running = True
        self._drop_c3 = False
        self._pair_bwd = True
        self.recompute = os.environ.get("MSFWSI_RECOMPUTE", "auto")  # off | t1 | targets | auto
        # halo-in-LDS 3x3 kernel: since the pure-DMA gather kernel lost its per-slab address arithmetic it only wins
        # for the 64-channel input gradient (494 vs 450 TFLOP/s); wider layers and the forward use the gather kernel
        self.halo3x3 = True
        self.fuse_pro3x3 = True  # 64->64 3x3: BatchNorm+ReLU in the conv's staging
        # Bottleneck conv3+bn3 backward folded into weights (no c3 in backward at all); 0 = keep / re-make c3
        self.fold_bn3 = True
        # ... and the closing ReLU gate of a folded block applied by the producer of its output gradient
        self.fuse_gate = True
        # forward of conv3 -> bn3 -> += identity -> relu in ONE conv launch: bn3's batch statistics come from the
        # Gram matrix of conv3's operand (sum c3 = W sum(a2), sum c3^2 = diag(W (a2^T a2) W^T)), c3 never exists
        self.fold_bn3_fwd = True
        # ... which also emits the block's closing ReLU gate as one byte per 16-byte chunk for the backward pass
        self.gate_bits = True
        self.fuse_two_source = True
        # ... whose second source is conv2's RAW output, normalised inside the launch (msfwsi_conv_dgrad2_pro): a2 is not
        # read there, and at 64 channels never stored
        self.dgrad2_pro = True
        # ... up to this width: at 64 channels (layer1, HBM-bound) the launch gets 13 % faster and a2 disappears (2.36 -> 2.06
        # ms per N = 4096 launch, profiles/r06_kbench_dgrad2pro.txt); from 128 channels on the launches are bound by L2 ingest
        # and issue slots, every column tile repeats the transform, and the saved read does not pay for it (1.19 -> 1.28 ms at
        # 28 x 28, 0.77 -> 0.95 at 14 x 14)
        self.dgrad2_pro_max_c = 64
        # activation-stationary ("panel") kernels for the short-k 1x1 convs of a Bottleneck (csrc/panel.hip): conv3's fused
        # tail reads conv2's RAW output (bn2 + ReLU applied while the panel is staged), conv1's input gradient forms bn1's
        # backward dc1 = k1*g + k2*c1 + k3 in its staging and writes it back once for the weight gradient
        self.multirank_streams = os.environ.get("MSFWSI_MULTIRANK_STREAMS", "1") != "0"
        self.panel_fwd = True
        self.panel_dgrad = True
        # conv2 of layer2 / layer3 on the image-stationary kernels (csrc/img3x3.hip): bn1 + ReLU in the forward staging, bn2's
        # backward in the gradient staging, a1 for the weight gradient as a by-product of the gradient's gate
        # the three head groups on the three streams of the multi-stream schedule (they are independent between the encoder
        # passes and the loss, and again between the loss and the encoder backward): their ~400 launches are too small to
        # fill the chip one after the other
        self.heads_on_streams = True
        self.img3x3 = True
        # layer1 (56x56x64): the weights-stationary kernel keeps the forward and the plain gradient (1.32 / 1.42 ms against 1.38
        # / 1.58 ms per N = 4096 launch); the image kernel takes only the gradient WITH bn2's backward folded in (2.04 against
        # 1.0 + 1.42 ms of msfwsi_bn_bwd_apply + gradient, profiles/r05_img3_kbench_c64.txt)
        self.img3x3_layer1 = True
        # the strided conv2 of layer2.0 / layer3.0: input gradient in ONE launch (msfwsi_img3x3_s2_dgrad) instead of four
        # parity launches, with bn2's backward and the a1 by-product as above
        self.img3x3_s2 = True
        self.gap_stride_fused = True  # gap_fwd + pixel_stride of a stage output in one pass
        # the folded tail's backward: a2 = relu(bn2(c2)) as a by-product of the M = g^T a2 launch (msfwsi_conv_wgrad_act)
        self.fuse_a2_wgrad = True
        self.fuse_a2_wgrad_max_c = 64
        self.img3x3_chunk_bytes = 1 << 30  # see _img3_bwd_chunks
        self.img3x3_min_fill = 1.0  # rounds of workgroups, see _img3_fills
        self._ncu: Dict[object, int] = {}
        self.panel_gram = True  # bn_act_sum + gram as ONE pass over the raw conv output
        self.panel_fwd_min_k = 128  # 56x56 / 64 channels: the gather kernel is at the HBM roof
        self.stem_run = True
        self.stem_s2d = True  # ... in space-to-depth form (4x4 / stride 1)
        self.stem_fuse_bnbwd = True  # bn1 backward inside the stem's dW
        self.pair_head_wgrad = True  # one dW launch for both views
        self.pair_head_fwd = True  # ... and one forward GEMM per layer
        # gradient exchange of the fuser heads in per-scale buckets, each launched when its weight gradients are complete
        self.bucket_inter = True
        # the heads' big Linear weight gradients (one launch per step for both views) stored instead of accumulated
        self.store_head_wgrad = True
        # stem backward as sums pass + apply pass (no gated gradient in memory): measured 2 ms SLOWER than
        # stem_pool_bwd + bn_bwd_apply (the pool-backward window logic is VALU-bound, not byte-bound): off
        self.fold_ds = True  # stride-1 downsample branch folded like bn3
        self.fold_ds_fwd = True  # ... and its forward: one two-source GEMM
        self.fold_ds_strided = True  # ... also for the stride-2 branches
        # their input gradient stays low-resolution: conv1's dgrad epilogue adds it on the strided sub-grid
        self.lores_resid = True
        self._stem_cache: Dict[tuple, tuple] = {}
        self._gate_vecs: Dict[Tuple[int, str], Tuple[torch.Tensor, torch.Tensor]] = {}
        self._plan_cache: Dict[tuple, Tuple[frozenset, bool]] = {}
        self._mode_override: Optional[str] = None  # "targets" while a model built with use_checkpoint=True runs
        self._msgs: Dict[tuple, torch.Tensor] = {}
        self._msg_lock = threading.Lock()
        # rehearsal switch: run the cross-replica code path (collectives included) even with one rank, so that the
        # RCCL calls of the SyncBatchNorm exchange execute on a one-GPU box
        self.force_sync = os.environ.get("MSFWSI_FORCE_SYNC", "0") != "0"
        # The two views of an encoder are independent passes (separate BatchNorm batches, backbone.py:140-145): view 1
        # runs on a second HIP stream beside view 0, forward and backward.  The passes are offset in time, so the
        # chip usually holds an MFMA-bound kernel of one beside an HBM-bound kernel of the other, and tails / launch
        # gaps of one are filled by the other.  Per BatchNorm module the running-statistics update of view 1 waits for
        # the one of view 0 (an event), which keeps the reference's update order.
        # Round 4: ON BY DEFAULT ON ONE RANK from the second step of a shape on, when the memory plan of the previous step
        # found room for a second set of transients (`_dual_ok`; the first step of a shape calibrates the plan on one
        # stream): 540.6 -> 523.3 ms/step on BASELINE config 2 (A/B on one box, two rounds, gpurun_out/r4_ab_dual.log), at
        # 257 GiB reserved instead of 233.5 (the side stream's allocator pool).  With more than one rank it stays off --
        # RCCL's buffers need that memory -- and the views run in LOCKSTEP instead (coalesced SyncBatchNorm messages,
        # _ViewPair).  MSFWSI_DUAL_STREAM=0 / 1 forces it off / on.  Per-kernel timings are those of two overlapping
        # streams when it is on.
        env = os.environ.get("MSFWSI_DUAL_STREAM")
        self.dual_stream: Optional[bool] = None if env is None else env != "0"
        self._dual_ok: Dict[tuple, bool] = {}
        self._dual_started: set = set()
        # the automatic multi-stream schedule needs an owner whose gradient accumulators exist before the backward starts
        # (the fused trainers: train.PretrainStep sets this); the process-wide default engine behind the plain model API
        # accumulates into lazily created buffers from autograd's thread and stays on one stream
        self.allow_multistream = False
        self.before_heads = None  # callable run on the launch stream right before the heads of a forward
        # ... and with two streams active, the two CONTEXT passes (1/17 of the images, launches too small to fill the chip:
        # 45 ms of a 545 ms step for 6 % of the work) run on a THIRD stream beside the target passes, forward and backward
        # (different encoder, different parameters: no ordering between them and the target passes).  Needs the
        # memory calibration of an earlier step of the same shape (`_calib`).  `ctx_stream = False` (MSFWSI_ENGINE=ctx_stream=0) turns it off.
        self.ctx_stream = True
        self._calib: Dict[tuple, Tuple[float, float]] = {}
        self._side: Dict[str, torch.cuda.Stream] = {}
        self._stream_groups: Dict[int, object] = {}  # side stream handle -> its own communicator (see _comm)
        self._bn_order: Optional[Tuple[str, dict]] = None
        # Cross-replica runs: the two views of an encoder in LOCKSTEP on two host threads, one SyncBatchNorm message per
        # BatchNorm and direction for both views (_ViewPair).  On whenever statistics are exchanged (more than one rank,
        # or MSFWSI_FORCE_SYNC) and the dual-stream schedule is off; `coalesce_views = False` restores one pass after the other
        self.coalesce_views = True
        self._tls = threading.local()   # .pair = (_ViewPair, view index) inside a lockstep pass
        # bookkeeping a trainer / bench.py reports: the collectives this engine issued since `reset_counters`, and
        # the recompute plan of the last forward ("keep-all", "recompute:t1", "recompute:t0,t1" [+ ",drop-c3"])
        self.collectives = 0
        self.collectives_last_step = 0
        self.last_plan = "keep-all"
        self.last_shape: Optional[tuple] = None
        self._apply_env_options()

    def reset_counters(self):
        self.collectives = 0

    def close_counters(self):
        self.collectives_last_step = self.collectives

    def invalidate_weights(self):
        """Every derived copy of the parameters (16-bit / channel-padded casts, the stem's filter-row runs) is keyed on
        torch's version counter and data pointer, which an optimizer that updates the storage through raw pointers
        (the flat Adam kernel) or `load_state_dict` into views does not move: trainers call this after every update."""
        self.weights.clear()
        self._stem_cache.clear()

    # ---- configuration ---------------------------------------------------------------------
    @staticmethod
    def compute_dtype() -> torch.dtype:
        env = os.environ.get("MSFWSI_DTYPE")
        if env:
            return {"fp32": torch.float32, "float32": torch.float32, "bf16": torch.bfloat16,
                    "bfloat16": torch.bfloat16, "fp16": torch.float16, "float16": torch.float16}[env.lower()]
        if torch.is_autocast_enabled():
            dt = torch.get_autocast_dtype("cuda")  # --amp: fp16 (reference default) or bf16 (--bf16)
            if dt in (torch.bfloat16, torch.float16):
                return dt
            raise _lib.MsfwsiHipError(f"autocast dtype {dt} is not implemented by the gfx950 kernels")
        return torch.float32

    def _world(self) -> int:
        if dist.is_available() and dist.is_initialized():
            return dist.get_world_size(self.group)
        return 1

    def _comm(self):
        """the process group the CURRENT stream's BatchNorm exchanges travel on.  Collectives of one communicator execute in
        the order they were enqueued: with all three streams on one communicator the second view's first exchange would
        queue behind ALL exchanges of the first view's pass -- the streams would run one after the other.  Each side stream
        gets its own communicator (created collectively, in a fixed order, before the first multi-stream step)."""
        if not self._stream_groups:
            return self.group
        return self._stream_groups.get(torch.cuda.current_stream().cuda_stream, self.group)

    def _make_stream_groups(self, dev):
        """COLLECTIVE (every rank, same order): communicators for the side streams of the multi-stream schedule"""
        if self._stream_groups or not (dist.is_available() and dist.is_initialized()):
            return
        backend = dist.get_backend(self.group)
        ranks = dist.get_process_group_ranks(self.group) if self.group is not None else None
        for which in ("side", "ctx"):
            g = dist.new_group(ranks=ranks, backend=backend)
            self._stream_groups[self._side_stream(dev, which).cuda_stream] = g
            probe_collectives(g, dev)  # the communicator's buffers exist before the memory plan measures what is free

    def prepare_multirank(self, dev):
        """COLLECTIVE, called once by the trainer's constructor when statistics are exchanged (more than one rank, or the
        single-rank RCCL rehearsal): the multi-stream schedule's side-stream communicators are created and probed HERE -- a
        point every rank passes in the same order -- and what they (and the main communicator's first collectives) took
        from the card is MEASURED (`comm_bytes`: torch.cuda.mem_get_info before / after; RCCL allocates its channel buffers
        and peer mappings outside torch's pool), so the memory plan's reserve is a measurement, not an assumption.  If a
        rank cannot create or use them the verdict is collective (MIN over ranks) and every rank runs the views in
        lockstep on the main communicator instead -- a slower schedule, never a dead first step."""
        if not (dist.is_available() and dist.is_initialized()):
            return
        backend = dist.get_backend(self.group)
        cdev = dev if backend == "nccl" else "cpu"
        torch.cuda.synchronize(dev)
        free0, _ = torch.cuda.mem_get_info(dev)
        probe_collectives(self.group, dev)
        ok = 1
        if self.multirank_streams and self.allow_multistream:
            try:
                self._make_stream_groups(dev)
            except Exception as e:  # noqa: BLE001 -- whatever the backend refuses is the finding
                print(f"[msf_wsi_amd] side-stream communicators unavailable ({type(e).__name__}: {e}); the views run in "
                      f"lockstep on the main communicator", flush=True)
                ok = 0
            flag = torch.full((1,), ok, dtype=torch.int32, device=cdev)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=self.group)
            if not int(flag[0]):
                self._stream_groups = {}
                self.multirank_streams = False
        torch.cuda.synchronize(dev)
        free1, _ = torch.cuda.mem_get_info(dev)
        self.comm_bytes = max(0, free0 - free1)

    comm_bytes = 0  # device memory the communicators took outside torch's pool (prepare_multirank)

    def _side_stream(self, dev, which: str = "side") -> "torch.cuda.Stream":
        key = f"{dev}/{which}"
        if key not in self._side:
            self._side[key] = torch.cuda.Stream(device=dev)
        return self._side[key]

    def _prewarm(self, encoders, dtype, dev):
        """derived weight copies and small constants that passes on BOTH streams read are made here, on the calling
        stream, before the second stream is released (a lazily filled cache would be written by one stream while the
        other already reads it)"""
        for enc in encoders:
            for m in enc.modules():
                if isinstance(m, (nn.Conv2d, nn.Linear)):
                    if m is getattr(enc, "conv1", None):
                        CP = chan_pad(dtype)
                        if self.stem_run:
                            self._stem_run_weights(m, dtype, CP)
                            if self.stem_s2d and m.kernel_size == (7, 7) and m.in_channels == 3:
                                self._stem_s2d_weights(m, dtype)
                        self.weights.get(m.weight, dtype, pad_to=CP)
                    else:
                        w = self.weights.get(m.weight, dtype)
                        self._unit_gate(m.weight.shape[0], dev)
                        if dtype != torch.float32 and isinstance(m, nn.Conv2d) and m.kernel_size == (1, 1):
                            self._f32_of(w)  # the fold algebra's fp32 copy (_gram_stats), shared by both streams
                            if m.stride == (1, 1) and m.bias is None:
                                self._panel_weights(m, w, dtype, dgrad=False)
                                self._panel_weights(m, w, dtype, dgrad=True)
                        if isinstance(m, nn.Conv2d):
                            self._img3_weights(m, w, dtype, dgrad=False)
                            self._img3_weights(m, w, dtype, dgrad=True)

    def _msg_buf(self, kind: str, n: int, dev) -> torch.Tensor:
        """pre-allocated fp64 message buffer of the cross-replica BatchNorm exchange, one per (direction, length).
        Re-use is safe: the blocking all-reduce makes the launch stream wait for the collective, and the consumer
        (bn_finalize / bn_bwd_finalize) precedes the next producer (shard_sum) in stream order."""
        key = (kind, n, str(dev), torch.cuda.current_stream(dev).cuda_stream)
        buf = self._msgs.get(key)
        if buf is None:
            buf = self._msgs[key] = torch.empty(n, dtype=torch.float64, device=dev)
        return buf

    def _pair_buf(self, kind: str, n: int, dev) -> torch.Tensor:
        """[2][n] fp64 message buffer of a lockstep exchange: row v is view v's packed vector, the collective sends both.
        Re-use across exchanges is safe for the reason given in _msg_buf; the two views only ever touch their own row."""
        key = (kind + "2", n, str(dev), torch.cuda.current_stream(dev).cuda_stream)
        buf = self._msgs.get(key)
        if buf is None:
            with self._msg_lock:
                buf = self._msgs.get(key)
                if buf is None:
                    buf = self._msgs[key] = torch.empty(2, n, dtype=torch.float64, device=dev)
        return buf

    def _sync(self, bn: nn.Module) -> bool:
        if self._world() == 1:
            return self.force_sync and dist.is_available() and dist.is_initialized()
        if self._sync_bn is not None:
            return self._sync_bn
        return isinstance(bn, nn.SyncBatchNorm)

    def _pair_collective(self, both: torch.Tensor):
        sync_sums(both.view(-1), self._comm(), force=self.force_sync)
        self.collectives += 1

    def _lockstep(self) -> bool:
        """the two views of an encoder run as a lockstep pair (see _ViewPair)"""
        if not self.coalesce_views:
            return False
        if self._world() > 1:
            return True
        return self.force_sync and dist.is_available() and dist.is_initialized()

    def _run_views(self, fn0, fn1):
        """run fn0() (view 0, on this thread) and fn1() (view 1, on a helper thread) as a lockstep pair on the CURRENT
        stream; returns (result0, result1).  The helper thread inherits nothing implicitly: device, stream and the no-grad
        mode are set explicitly (they are thread-local in torch)."""
        pair = _ViewPair()
        dev = torch.cuda.current_device()
        stream = torch.cuda.current_stream(dev)
        res: List[object] = [None, None]
        err: List[Optional[BaseException]] = [None, None]

        def body(v, fn):
            self._tls.pair = (pair, v)
            try:
                res[v] = fn()
            except BaseException as e:  # noqa: BLE001 -- handed to the calling thread
                err[v] = e
                pair.fail(e)
            finally:
                self._tls.pair = None

        def helper():
            with torch.cuda.device(dev), torch.cuda.stream(stream), torch.no_grad():
                body(1, fn1)

        t = threading.Thread(target=helper, name="msfwsi-view1", daemon=True)
        t.start()
        body(0, fn0)
        t.join()
        for e in err:
            if e is not None and not (isinstance(e, RuntimeError) and "other view pass" in str(e)):
                raise e
        for e in err:
            if e is not None:
                raise e
        return res[0], res[1]

    # ---- BatchNorm helpers ---------------------------------------------------------------------
    @staticmethod
    def _bn_frozen(bn: nn.Module) -> bool:
        """eval-mode BatchNorm with running statistics (model.eval(), nn.BatchNorm*: `not training and
        track_running_stats`): normalises with running_mean / running_var, nothing is reduced or exchanged"""
        return (not bn.training) and bn.track_running_stats and bn.running_mean is not None

    def _bn_eval_state(self, bn: nn.Module, count: int) -> BNState:
        Cn = bn.num_features
        vecs = torch.empty(4, Cn, dtype=torch.float32, device=bn.running_mean.device)
        kn.bn_eval_coeffs(bn.running_mean, bn.running_var, bn.weight if bn.affine else None,
                          bn.bias if bn.affine else None, bn.eps, vecs[0], vecs[1], vecs[2], vecs[3])
        return BNState(vecs[0], vecs[1], vecs[2], vecs[3], float(count), frozen=True)

    def _bn_finalize(self, stats: torch.Tensor, count: int, bn: nn.Module) -> BNState:
        if self._bn_frozen(bn):
            return self._bn_eval_state(bn, count)
        Cn = stats.shape[-1]
        dev = stats.device
        vecs = torch.empty(4, Cn, dtype=torch.float32, device=dev)
        total = float(count)
        pv = getattr(self._tls, "pair", None)
        if self._sync(bn):
            if pv is not None:  # lockstep: both views' [sum, sumsq] in ONE message
                pair, v = pv
                both = self._pair_buf("fwd", 2 * Cn, dev)
                kn.shard_sum(stats, both[v])
                pair.exchange(v, lambda: self._pair_collective(both))
                packed = both[v]
            else:
                packed = self._msg_buf("fwd", 2 * Cn, dev)
                kn.shard_sum(stats, packed)
                sync_sums(packed, self._comm(), force=self.force_sync)  # RCCL sum of [sum, sumsq]; equal shards per rank
                self.collectives += 1
            stats = packed.view(1, 2, Cn)
            total *= self._world()
        if bn.momentum is None:
            raise NotImplementedError("cumulative-average BatchNorm (momentum=None) is not used by MSF-WSI")
        track = self.update_running and bn.track_running_stats and bn.training
        order = self._bn_order if track else None
        if order is not None and order[0] == "follow":
            ev = order[1].get(id(bn))
            if ev is not None:  # view 0's update of this module's running statistics comes first (reference order)
                torch.cuda.current_stream(dev).wait_event(ev)
        if pv is not None and track:
            pv[0].turn_wait(pv[1])  # view 0's update of this module's running statistics is enqueued first
        kn.bn_finalize(stats, total, bn.weight if bn.affine else None, bn.bias if bn.affine else None, bn.eps,
                       bn.momentum, bn.running_mean if track else None, bn.running_var if track else None,
                       bn.num_batches_tracked if track else None, vecs[0], vecs[1], vecs[2], vecs[3])
        if pv is not None and track:
            pv[0].turn_done(pv[1])
        if order is not None and order[0] == "lead":
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(dev))
            order[1][id(bn)] = ev
        return BNState(vecs[0], vecs[1], vecs[2], vecs[3], total)

    def _bn_bwd_coeffs(self, sums: torch.Tensor, nslots: int, which: int, bn: nn.Module, st: BNState,
                       grads: GradStore):
        if st.frozen:
            raise NotImplementedError(
                "backward through an eval-mode BatchNorm (frozen running statistics) is not implemented by the "
                "MSF-WSI HIP engine: call model.train() for training, or run eval-mode forwards under torch.no_grad()")
        Cn = sums.shape[-1]
        dev = sums.device
        if self._sync(bn):
            pv = getattr(self._tls, "pair", None)
            if pv is not None:
                both = self._pair_buf("bwd", nslots * Cn, dev)
                packed = both[pv[1]]
            else:
                packed = self._msg_buf("bwd", nslots * Cn, dev)
            kn.shard_sum(sums, packed)
            k = torch.empty(3, Cn, dtype=torch.float32, device=dev)
            if bn.affine:
                # parameter gradients are averaged over replicas later (DDP): dgamma/dbeta come from the LOCAL sums,
                # taken from the message buffer before the exchange overwrites it (k is scratch here)
                kn.bn_bwd_finalize(packed.view(1, nslots, Cn), nslots, which, st.count, bn.weight, st.mean, st.invstd,
                                   grads.get(bn.weight), grads.get(bn.bias), k[0], k[1], k[2])
            if pv is not None:  # lockstep: one message for both views (enqueued after both local finalizes)
                pv[0].exchange(pv[1], lambda: self._pair_collective(both))
            else:
                sync_sums(packed, self._comm(), force=self.force_sync)
                self.collectives += 1
            kn.bn_bwd_finalize(packed.view(1, nslots, Cn), nslots, which, st.count,
                               bn.weight if bn.affine else None, st.mean, st.invstd, None, None, k[0], k[1], k[2])
            return k
        k = torch.empty(3, Cn, dtype=torch.float32, device=dev)
        kn.bn_bwd_finalize(sums, nslots, which, st.count, bn.weight if bn.affine else None, st.mean, st.invstd,
                           grads.get(bn.weight) if bn.affine else None, grads.get(bn.bias) if bn.affine else None,
                           k[0], k[1], k[2])
        return k

    def _bn_finalize_views(self, stats2: Sequence[torch.Tensor], counts: Sequence[int], bn: nn.Module) -> List[BNState]:
        """_bn_finalize for the two views of a head at once (both statistics tensors are at hand on one thread): with
        cross-replica statistics ONE message carries both views' [sum, sumsq]; view 0's running-statistics update is
        enqueued first (backbone.py:161-186 calls the head on view 1 first and view 2 second)"""
        if not (self._sync(bn) and self._lockstep()) or self._bn_frozen(bn):
            return [self._bn_finalize(st, c, bn) for st, c in zip(stats2, counts)]
        Cn = stats2[0].shape[-1]
        dev = stats2[0].device
        both = self._pair_buf("fwd", 2 * Cn, dev)
        for v in range(2):
            kn.shard_sum(stats2[v], both[v])
        self._pair_collective(both)
        if bn.momentum is None:
            raise NotImplementedError("cumulative-average BatchNorm (momentum=None) is not used by MSF-WSI")
        track = self.update_running and bn.track_running_stats and bn.training
        out = []
        for v in range(2):
            vecs = torch.empty(4, Cn, dtype=torch.float32, device=dev)
            total = float(counts[v]) * self._world()
            kn.bn_finalize(both[v].view(1, 2, Cn), total, bn.weight if bn.affine else None,
                           bn.bias if bn.affine else None, bn.eps, bn.momentum, bn.running_mean if track else None,
                           bn.running_var if track else None, bn.num_batches_tracked if track else None,
                           vecs[0], vecs[1], vecs[2], vecs[3])
            out.append(BNState(vecs[0], vecs[1], vecs[2], vecs[3], total))
        return out

    def _bn_bwd_coeffs_views(self, sums2: Sequence[torch.Tensor], nslots: int, which: int, bn: nn.Module,
                             sts: Sequence[BNState], grads: GradStore) -> List[torch.Tensor]:
        """_bn_bwd_coeffs for the two views of a head at once: one message for both views' [sum g, sum g*c]"""
        if not (self._sync(bn) and self._lockstep()) or any(st.frozen for st in sts):
            return [self._bn_bwd_coeffs(sm, nslots, which, bn, st, grads) for sm, st in zip(sums2, sts)]
        Cn = sums2[0].shape[-1]
        dev = sums2[0].device
        both = self._pair_buf("bwd", nslots * Cn, dev)
        ks = []
        for v in range(2):
            kn.shard_sum(sums2[v], both[v])
            k = torch.empty(3, Cn, dtype=torch.float32, device=dev)
            if bn.affine:  # dgamma / dbeta from the LOCAL sums, before the exchange overwrites them
                kn.bn_bwd_finalize(both[v].view(1, nslots, Cn), nslots, which, sts[v].count, bn.weight, sts[v].mean,
                                   sts[v].invstd, grads.get(bn.weight), grads.get(bn.bias), k[0], k[1], k[2])
            ks.append(k)
        self._pair_collective(both)
        for v in range(2):
            kn.bn_bwd_finalize(both[v].view(1, nslots, Cn), nslots, which, sts[v].count,
                               bn.weight if bn.affine else None, sts[v].mean, sts[v].invstd, None, None,
                               ks[v][0], ks[v][1], ks[v][2])
        return ks

    # ---- single conv / linear unit ---------------------------------------------------------------
    def _unit_fwd(self, op: nn.Module, bn: Optional[nn.Module], relu: bool, x: torch.Tensor,
                  x_pro: Optional[BNState], geom, dtype: torch.dtype, pad_c: int = 0) -> Unit:
        N, H, W, Cin = geom
        if isinstance(op, nn.Conv2d):
            K, R, S = op.out_channels, op.kernel_size[0], op.kernel_size[1]
            stride, pad = op.stride[0], op.padding[0]
        else:
            K, R, S, stride, pad = op.out_features, 1, 1, 1, 0
        d = kn.conv_desc(dtype, N, H, W, Cin, K, R, S, stride, pad)
        w = self.weights.get(op.weight, dtype, pad_to=pad_c)
        c = torch.empty(N, d.P, d.Q, K, dtype=dtype, device=x.device)
        stats = kn.new_stats(K, 2, x.device) if bn is not None else None
        # BatchNorm1d of the heads: pooled features of different tiles are nearly equal, so a Linear output's batch mean
        # is 10-100x its batch deviation and E[c^2] - mean^2 cancels 3-4 digits; the GEMM epilogue's fp32 partial sums
        # are not enough there (1e-4 on invstd) -> statistics by a separate fp64 pass over the (small) output
        epi_stats = stats if not isinstance(op, nn.Linear) else None
        bias = getattr(op, "bias", None)
        xin, pro = x, (x_pro.scale, x_pro.shift) if x_pro is not None else None
        fuse_pro = (pro is not None and self.fuse_pro3x3 and bias is None and not pad_c and kn.conv3x3_stationary(d))
        wimg = (self._img3_weights(op, w, dtype, dgrad=False)
                if not pad_c and kn.img3x3_supported(d) and self._img3_fills(d, x.device) else None)
        if wimg is not None:
            # image-stationary kernel: the band is staged once, BatchNorm + ReLU of the producer applied on the way
            if not kn.img3x3_fwd(d, x, wimg, c, stats=stats, pro=pro):
                raise _lib.MsfwsiHipError("img3x3_fwd refused a geometry msfwsi_img3x3_supported accepted")
            u = Unit(op, bn, relu, d, x, x_pro, c)
            if bn is not None:
                u.st = self._bn_finalize(stats, N * d.P * d.Q, bn)
            return u
        if fuse_pro:
            pass  # the weights-stationary 3x3 kernel applies BatchNorm + ReLU on the way into LDS: nothing to materialise
        elif pro is not None:
            # a 3x3 gather reads every input element 9 times: normalising it once into a transient tensor and
            # letting the conv stage by pure LDS-DMA is cheaper than re-applying BatchNorm+ReLU per tap
            xin = torch.empty_like(x)
            kn.bn_act(x, pro[0], pro[1], xin, relu=True)
            pro = None
        if pad_c and pro is None and bias is None and self.stem_run and self._stem_run_fwd(op, xin, c, stats, dtype, pad_c):
            pass  # stem: 7 row taps over runs of contiguous pixels on the pure-DMA kernel
        elif fuse_pro:
            kn.conv3x3_fwd(d, xin, w, c, stats=stats, pro=pro)
        elif (pro is None and bias is None and not pad_c
              and kn.conv3x3_stationary(d)):
            kn.conv3x3_fwd(d, xin, w, c, stats=stats)  # input patch staged once per channel slab, 9 taps reuse it
        else:
            kn.conv_fwd(d, xin, w, c, pro=pro, bias=bias.data if bias is not None else None, stats=epi_stats)
            if stats is not None and epi_stats is None:
                kn.colstats(c, stats)
        u = Unit(op, bn, relu, d, x, x_pro, c)
        if bn is not None:
            u.st = self._bn_finalize(stats, N * d.P * d.Q, bn)
        return u

    def _stem_run_fwd(self, op: nn.Module, x: torch.Tensor, c: torch.Tensor, stats, dtype, CP: int) -> bool:
        """the stem conv as filter-row runs; False when the library has no run kernel for the shape"""
        R, S = op.kernel_size
        return kn.stem_conv_fwd(x, self._stem_run_weights(op, dtype, CP), c, stats, R, S, op.stride[0], op.padding[0])

    # ---- stem in space-to-depth form ---------------------------------------------------------------
    def _stem_s2d_ok(self, op: nn.Module, H: int, W: int) -> bool:
        return (self.stem_s2d and self.stem_run and isinstance(op, nn.Conv2d) and op.kernel_size == (7, 7)
                and op.stride == (2, 2) and op.padding == (3, 3) and op.in_channels == 3 and op.out_channels == 64
                and op.bias is None and H % 2 == 0 and W % 2 == 0)

    def _stem_s2d_weights(self, op: nn.Module, dtype) -> torch.Tensor:
        """[K][7][7][3] -> [K][4][4][16] (DESIGN 3.2 / msfwsi_stem_s2d_weights), cached per parameter version"""
        key = (id(op.weight), dtype, "s2d")
        ver = (op.weight._version, op.weight.data_ptr())
        hit = self._stem_cache.get(key)
        if hit is None or hit[0] != ver or hit[2]() is not op.weight:
            w2 = torch.empty(op.out_channels, 4, 4, 16, dtype=dtype, device=op.weight.device)
            kn.stem_s2d_weights(WeightStore.physical(op.weight), w2)
            hit = (ver, w2, weakref.ref(op.weight))
            self._stem_cache[key] = hit
        return hit[1]

    def _stem_s2d_fwd(self, enc: nn.Module, x: torch.Tensor, dtype) -> Optional[Unit]:
        """conv1 (7x7 / stride 2 / pad 3, resnet.py:174) as a 4x4 / stride-1 conv on the space-to-depth input: k range
        256 instead of 448 (7 row taps x 64-element runs of 8-padded channels), input tensor half the bytes"""
        op, bn = enc.conv1, enc.bn1
        N, _, H, W = x.shape
        H2, W2 = H // 2, W // 2
        xs = torch.empty(N, H2, W2, 16, dtype=dtype, device=x.device)
        kn.nchw_to_s2d(x, xs)
        c = torch.empty(N, H2, W2, op.out_channels, dtype=dtype, device=x.device)
        stats = kn.new_stats(op.out_channels, 2, x.device)
        if not kn.stem_conv_fwd(xs, self._stem_s2d_weights(op, dtype), c, stats, 4, 4, 1, 2, P=H2, Q=W2):
            return None
        d = _lib.ConvDesc(kn.dt_of(xs), N, H2, W2, 16, H2, W2, op.out_channels, 4, 4, 1, 2)
        u = Unit(op, bn, True, d, xs, None, c)
        u.st = self._bn_finalize(stats, N * H2 * W2, bn)
        u.s2d = True
        return u

    def _stem_run_weights(self, op: nn.Module, dtype, CP: int) -> torch.Tensor:
        """stem weights [K][R][S][CP] -> [K][R][run] (zero columns pad the S*CP run to whole k slabs), cached per
        parameter version"""
        K, R, S = op.out_channels, op.kernel_size[0], op.kernel_size[1]
        bk = 16 if dtype == torch.float32 else 32
        run = (S * CP + bk - 1) // bk * bk
        key = (id(op.weight), dtype, "run")
        ver = (op.weight._version, op.weight.data_ptr())
        hit = self._stem_cache.get(key)
        if hit is None or hit[0] != ver or hit[2]() is not op.weight:
            phys = WeightStore.physical(op.weight)  # fp32 [K][R][S][Cin]
            Cin = phys.shape[-1]
            wp = torch.empty(K * R * S, CP, dtype=torch.float32, device=phys.device)
            kn.pad_cast(phys, wp, K * R * S, Cin, CP)          # channels -> one 16-byte chunk
            w_run = torch.empty(K, R, run, dtype=dtype, device=phys.device)
            kn.pad_cast(wp, w_run, K * R, S * CP, run)         # filter row -> whole k slabs, storage type
            hit = (ver, w_run, weakref.ref(op.weight))
            self._stem_cache[key] = hit
        return hit[1]

    def _gram_stats(self, w: torch.Tensor, A: torch.Tensor, sa: torch.Tensor, bn: nn.Module, count: int, dtype) -> BNState:
        """BatchNorm batch statistics of c = W a (1x1 conv) from the Gram matrix A = a^T a and the column sums of a:
        sum c = W sum(a), sum c^2 = diag(W A W^T); w = the compute-dtype weights [K][1][1][C] the MFMA multiplies"""
        if self._bn_frozen(bn):
            return self._bn_eval_state(bn, count)
        K, Cw = w.shape[0], w.shape[-1]
        dev = w.device
        Wq = w if dtype == torch.float32 else self._f32_of(w)
        dlin = kn.conv_desc(torch.float32, K, 1, 1, Cw, Cw, 1, 1, 1, 0)
        WA = torch.empty(K, 1, 1, Cw, dtype=torch.float32, device=dev)
        kn.conv_fwd(dlin, Wq, A, WA)  # A is symmetric
        stats = kn.zeros((1, 2, K), torch.float64, dev)
        kn.fold_matvec(Wq, sa, stats[0, 0])
        kn.fold_dots(Wq, WA, stats[0, 1])
        return self._bn_finalize(stats, count, bn)

    def _panel_weights(self, op: nn.Module, w: torch.Tensor, dtype, dgrad: bool) -> Optional[torch.Tensor]:
        """the 1x1 conv weight [K][1][1][C] in MFMA fragment order for the panel kernels (msfwsi_panel_pack_weights), shared
        by the passes of a step (dropped by invalidate_weights); None where the panel kernels do not serve the layer"""
        if dtype == torch.float32 or not isinstance(op, nn.Conv2d):
            return None
        K, Cn = op.out_channels, op.in_channels
        nout, kk = (Cn, K) if dgrad else (K, Cn)
        if kk not in (64, 128, 256, 512) or nout < 128 or nout % 32:
            return None
        if dgrad:  # the forward tensor read as the [k = K][n = C] operand
            return self.weights.derived("panel_dgrad", w, lambda t: kn.panel_pack_weights(t, torch.empty_like(t), Cn, K, 1, Cn))
        return self.weights.derived("panel_fwd", w, lambda t: kn.panel_pack_weights(t, torch.empty_like(t), K, Cn, Cn, 1))

    def _img3_weights(self, op: nn.Module, w: torch.Tensor, dtype, dgrad: bool) -> Optional[torch.Tensor]:
        """a 3x3 / stride 1 conv weight [K][3][3][C] of the widths the image-stationary kernels serve, in the fragment order
        they stream (msfwsi_img3x3_pack_weights; the gradient's copy transposed with flipped taps); shared by the passes of
        a step like _panel_weights.  None for every other layer."""
        if (not self.img3x3 or dtype == torch.float32 or not isinstance(op, nn.Conv2d) or op.kernel_size != (3, 3)
                or op.padding != (1, 1) or op.groups != 1 or op.bias is not None or op.in_channels != op.out_channels):
            return None
        if op.stride == (2, 2):  # conv2 of layer2.0 / layer3.0: the input gradient only, in the passes' tap order
            if not (dgrad and self.img3x3_s2 and op.in_channels in (128, 256)):
                return None
            return self.weights.derived("img3_s2d", w, lambda t: kn.img3x3_pack_weights(t, torch.empty_like(t), 2))
        if (op.stride != (1, 1) or op.in_channels not in (128, 256, 64)
                or (op.in_channels == 64 and not (dgrad and self.img3x3_layer1))):
            return None
        return self.weights.derived("img3_dgrad" if dgrad else "img3_fwd", w,
                                    lambda t: kn.img3x3_pack_weights(t, torch.empty_like(t), dgrad))

    def _f32_of(self, w16: torch.Tensor) -> torch.Tensor:
        """fp32 copy of a 16-bit weight tensor, shared by the passes of one step (dropped by invalidate_weights)"""
        return self.weights.derived("f32", w16, kn.upcast_f32)

    def _ds_tail_fwd(self, conv3, bn3, dconv, dbn, c2: torch.Tensor, pro: BNState, x: torch.Tensor, geom, dtype,
                     want_bits: bool):
        """y = relu(bn3(conv3(a2)) + bn_d(conv_d(x))) for a Bottleneck with a stride-1 downsample branch
        (layer1.0, resnet.py:131-138 with self.downsample) as ONE two-source GEMM: both BatchNorms' batch statistics
        come from Gram matrices (of a2 and of x), their scales are folded into the weight rows and their shifts
        summed; neither conv output exists.  Returns (unit3, unit_ds, y, gate bits)."""
        N, H, W, Cw = geom
        K, Ci = conv3.out_channels, x.shape[-1]
        dev = c2.device
        d3 = kn.conv_desc(dtype, N, H, W, Cw, K, 1, 1, 1, 0)
        dd = kn.conv_desc(dtype, N, H, W, Ci, K, 1, 1, 1, 0)
        a2 = torch.empty_like(c2)
        A2 = sa = Ax = sx = None
        if self._bn_frozen(bn3):  # eval mode: statistics are the running ones, no Gram matrix needed
            kn.bn_act(c2, pro.scale, pro.shift, a2, relu=True)
        else:
            sa = kn.zeros((Cw,), torch.float64, dev)
            kn.bn_act_sum(c2, pro.scale, pro.shift, a2, sa)
            A2 = kn.zeros((Cw, 1, 1, Cw), torch.float32, dev)
            kn.gram(kn.conv_desc(dtype, N, H, W, Cw, Cw, 1, 1, 1, 0), a2, A2)
        if not self._bn_frozen(dbn):
            Ax = kn.zeros((Ci, 1, 1, Ci), torch.float32, dev)
            kn.gram(kn.conv_desc(dtype, N, H, W, Ci, Ci, 1, 1, 1, 0), x, Ax)
            sx = kn.zeros((Ci,), torch.float64, dev)
            kn.colsum(x, sx)
        st3 = self._gram_stats(self.weights.get(conv3.weight, dtype), A2, sa, bn3, N * H * W, dtype)
        std = self._gram_stats(self.weights.get(dconv.weight, dtype), Ax, sx, dbn, N * H * W, dtype)
        wcat32 = torch.empty(K, Cw + Ci, dtype=torch.float32, device=dev)
        shift = torch.empty(K, dtype=torch.float32, device=dev)
        kn.row_scale_cat(WeightStore.physical(conv3.weight), st3.scale, WeightStore.physical(dconv.weight), std.scale,
                         st3.shift, std.shift, wcat32, shift)
        wcat = wcat32 if dtype == torch.float32 else kn.cast_lowp(wcat32, torch.empty_like(wcat32, dtype=dtype))
        one, _ = self._unit_gate(K, dev)
        y = torch.empty(N, H, W, K, dtype=dtype, device=dev)
        bits = kn.gate_bytes(N * H * W, K, dtype, dev) if want_bits else None
        if not kn.conv_fwd_post2(d3, a2, wcat, y, x, one, shift, relu=True, gate_out=bits):
            return None
        u3 = Unit(conv3, bn3, False, d3, c2, pro, None, st3, gram=(A2, sa) if A2 is not None else None)
        ud = Unit(dconv, dbn, False, dd, x, None, None, std, gram=(Ax, sx) if Ax is not None else None)
        return u3, ud, y, bits

    def _conv_bn_res_fwd(self, conv: nn.Module, bn: nn.Module, c_in: torch.Tensor, pro: BNState, ident: torch.Tensor,
                         geom, dtype: torch.dtype, want_bits: bool = False):
        """y = relu(bn(conv1x1(a)) + ident), a = relu(pro(c_in)), without the conv output c = W a ever reaching HBM
        (src/models/resnet.py:131-138 for a Bottleneck without downsample).  BatchNorm's batch statistics are
        quadratic in c and follow from the operand alone:
            sum_p c[k]   = W[k,:] . sum_p a            sum_p c[k]^2 = W[k,:] (a^T a) W[k,:]^T
        so: one pass over a for its Gram matrix (a weight-gradient launch, 1/4 of the conv's FLOPs), tiny [K][C]
        algebra, then the conv with BatchNorm apply + residual + ReLU in its epilogue.  The Gram matrix and the
        column sums are kept: the folded backward (_block_end_folded) needs exactly them."""
        N, H, W, Cw = geom
        K = conv.out_channels
        dev = c_in.device
        d = kn.conv_desc(dtype, N, H, W, Cw, K, 1, 1, 1, 0)
        w = self.weights.get(conv.weight, dtype)
        # panel kernel (csrc/panel.hip): conv3 reads conv2's RAW output, bn2 + ReLU are applied while its operand panel is
        # staged; with the fused Gram pass the normalised activation a2 is never written at all
        wpk = None
        if self.panel_fwd and kn.panel_supported(d, False):
            wpk = self._panel_weights(conv, w, dtype, dgrad=False)
        a = A = sa = None
        if self._bn_frozen(bn):  # eval mode: statistics are the running ones, no Gram matrix needed
            if wpk is None:
                a = torch.empty_like(c_in)
                kn.bn_act(c_in, pro.scale, pro.shift, a, relu=True)
        else:
            sa = kn.zeros((Cw,), torch.float64, dev)
            A = kn.zeros((Cw, 1, 1, Cw), torch.float32, dev)
            if not (wpk is not None and self.panel_gram and kn.panel_gram(c_in, pro.scale, pro.shift, A, sa)):
                a = torch.empty_like(c_in)
                kn.bn_act_sum(c_in, pro.scale, pro.shift, a, sa)
                kn.gram(kn.conv_desc(dtype, N, H, W, Cw, Cw, 1, 1, 1, 0), a, A)
                if Cw < self.panel_fwd_min_k:
                    wpk = None  # a2 exists anyway and the gather kernel is at the HBM roof for these widths
        st = self._gram_stats(w, A, sa, bn, N * H * W, dtype)
        y = torch.empty(N, H, W, K, dtype=dtype, device=dev)
        bits = kn.gate_bytes(N * H * W, K, dtype, dev) if want_bits else None
        if wpk is not None:
            if not kn.panel_fwd_post(d, c_in, wpk, y, st.scale, st.shift, pro=(pro.scale, pro.shift), ident=ident, relu=True,
                                     gate_out=bits):
                raise _lib.MsfwsiHipError("panel_fwd_post refused a geometry msfwsi_panel_supported accepted")
        else:
            kn.conv_fwd_post(d, a, w, y, st.scale, st.shift, ident=ident, relu=True, gate_out=bits)
        u = Unit(conv, bn, False, d, c_in, pro, None, st, gram=(A, sa) if A is not None else None)
        return u, y, bits

    def _normalised_operand(self, u: Unit) -> torch.Tensor:
        """relu(bn(x)) of a unit's operand as a transient tensor (one streaming pass)"""
        xm = torch.empty_like(u.x)
        kn.bn_act(u.x, u.x_pro.scale, u.x_pro.shift, xm, relu=True)
        return xm

    def _unit_wgrad(self, u: Unit, dc: torch.Tensor, grads: GradStore, dtype: torch.dtype, x_mat=None):
        pro = (u.x_pro.scale, u.x_pro.shift) if u.x_pro is not None else None
        x = u.x
        if x_mat is not None:
            x, pro = x_mat, None
        elif pro is not None and self.fuse_pro3x3 and kn.conv_wgrad_stationary(u.desc) and not u.s2d:
            pass  # the output-stationary kernel applies BatchNorm + ReLU in its staging: nothing to materialise
        elif pro is not None and u.desc.N * u.desc.H * u.desc.W >= 8192:
            # normalise the operand once into a transient tensor: the weight-gradient kernel then stages both
            # tiles by LDS-DMA (3-stage pipeline) instead of register-staging with the BatchNorm prologue
            x = torch.empty_like(u.x)
            kn.bn_act(u.x, pro[0], pro[1], x, relu=True)
            pro = None
        if u.s2d:  # weight gradient on the space-to-depth operand, folded back into [K][7][7][3]
            dw2 = kn.zeros((u.desc.K, 4, 4, 16), torch.float32, dc.device)
            kn.conv_wgrad(u.desc, u.x, dc, dw2)
            kn.stem_s2d_wfold(dw2, grads.get(u.op.weight))
        elif u.desc.C != u.op.weight.shape[1]:  # channel-padded stem
            CP = u.desc.C
            dwp = torch.zeros(u.desc.K, u.desc.R, u.desc.S, CP, dtype=torch.float32, device=dc.device)
            kn.conv_wgrad(u.desc, x, dc, dwp, pro=pro)
            rows = u.desc.K * u.desc.R * u.desc.S
            kn.unpad_add(dwp, grads.get(u.op.weight), rows, u.op.weight.shape[1], CP)
        else:
            kn.conv_wgrad(u.desc, x, dc, grads.get(u.op.weight), pro=pro)
        bias = getattr(u.op, "bias", None)
        if bias is not None:
            cs = torch.zeros(u.desc.K, dtype=torch.float64, device=dc.device)
            kn.colsum(dc, cs)
            kn.add_f64_to_f32(cs, grads.get(bias), 1.0)

    def _unit_dgrad(self, u: Unit, dc: torch.Tensor, dtype: torch.dtype, resid=None, gapg=None, gap_scale=0.0,
                    mask=None, sums=None, mask_bits=None, resid_stride: int = 1):
        d = u.desc
        dx = torch.empty(d.N, d.H, d.W, d.C, dtype=dtype, device=dc.device)
        w = self.weights.get(u.op.weight, dtype)
        if (self.halo3x3 and gapg is None and mask_bits is None and resid_stride == 1 and d.K <= 64
                and kn.conv3x3_supported(d)):
            kn.conv3x3_dgrad(d, dc, w, dx, resid=resid, mask=mask, sums=sums)
        else:
            kn.conv_dgrad(d, dc, w, dx, resid=resid, gapg=gapg, gap_scale=gap_scale, mask=mask, sums=sums,
                          mask_bits=mask_bits, resid_stride=resid_stride)
        return dx

    # ---- encoder -------------------------------------------------------------------------------
    def encoder_forward(self, enc: nn.Module, x: torch.Tensor, dtype: torch.dtype, save: bool = True) -> EncPass:
        """save=False: features only; activations are dropped block by block and the pass is recomputed
        right before its backward (pass-level recompute, the engine's answer to the reference's --use-ac)."""
        if not x.is_cuda:
            raise _lib.MsfwsiHipError("the MSF-WSI encoders run only on a HIP device (no CPU path)")
        if x.dim() != 4 or x.shape[1] != 3:
            raise ValueError(f"expected [N,3,H,W] images, got {tuple(x.shape)}")
        x = x.detach()
        if x.dtype != torch.float32:
            x = x.float()
        x = x.contiguous()
        N, _, H, W = x.shape
        CP = chan_pad(dtype)
        stem = self._stem_s2d_fwd(enc, x, dtype) if self._stem_s2d_ok(enc.conv1, H, W) else None
        if stem is not None:
            xin = stem.x
        else:
            xin = torch.empty(N, H, W, CP, dtype=dtype, device=x.device)
            kn.nchw_to_nhwc(x, xin, CP)
            stem = self._unit_fwd(enc.conv1, enc.bn1, True, xin, None, (N, H, W, CP), dtype, pad_c=CP)
        H0, W0 = stem.desc.P, stem.desc.Q
        P, Q = (H0 - 1) // 2 + 1, (W0 - 1) // 2 + 1
        pooled = torch.empty(N, P, Q, 64, dtype=dtype, device=x.device)
        amax = torch.empty(N, P, Q, 64, dtype=torch.uint8, device=x.device)
        kn.stem_pool_fwd(stem.c, stem.st.scale, stem.st.shift, pooled, amax, N, H0, W0, 64)
        y, h, w = pooled, P, Q
        blocks: List[BlockRec] = []
        feats: List[torch.Tensor] = []
        stages_list = list(enc.stages())
        strided_next = None  # y[:, ::2, ::2, :] of the previous stage's output, when its pooling pass made it
        for si, stage in enumerate(stages_list):
            nb = len(stage)
            for bi, blk in enumerate(stage):
                cin = y.shape[-1]
                units: List[Unit] = []
                cur, cur_pro, gh, gw = y, None, h, w
                main = blk.main_branch()
                conv3 = main[-1][0]
                fused_tail = (self.fold_bn3_fwd and len(main) == 3 and blk.downsample is None
                              and conv3.kernel_size == (1, 1) and conv3.stride == (1, 1) and conv3.bias is None)
                dsc = blk.downsample[0] if blk.downsample is not None else None
                bkk = 16 if dtype == torch.float32 else 32
                ds_tail = (self.fold_bn3_fwd and self.fold_bn3 and self.fold_ds and self.fold_ds_fwd and len(main) == 3
                           and dsc is not None and conv3.kernel_size == (1, 1) and conv3.stride == (1, 1)
                           and conv3.bias is None and dsc.kernel_size == (1, 1) and dsc.stride in ((1, 1), (2, 2))
                           and dsc.padding == (0, 0) and dsc.bias is None and dsc.in_channels % bkk == 0
                           and conv3.in_channels % bkk == 0 and main[1][0].stride == dsc.stride
                           and main[0][0].stride == (1, 1) and (dsc.stride == (1, 1) or self.fold_ds_strided))
                for ui, (conv, bn) in enumerate(main[:-1] if (fused_tail or ds_tail) else main):
                    u = self._unit_fwd(conv, bn, ui + 1 < len(main), cur, cur_pro, (N, gh, gw, cur.shape[-1]), dtype)
                    units.append(u)
                    cur, cur_pro, gh, gw = u.c, u.st, u.desc.P, u.desc.Q
                ds = None
                if ds_tail:
                    xs = y
                    if dsc.stride[0] > 1:  # the branch's operand as a dense tensor: y[:, ::s, ::s, :]
                        xs = strided_next  # made by the previous stage's pooling pass (gap_fwd_stride2), if it could
                        strided_next = None
                        if xs is None or tuple(xs.shape) != (N, gh, gw, cin):
                            xs = torch.empty(N, gh, gw, cin, dtype=dtype, device=y.device)
                            kn.pixel_stride(y, xs, dsc.stride[0], expand=False)
                    got = self._ds_tail_fwd(conv3, main[-1][1], blk.downsample[0], blk.downsample[1], cur, cur_pro, xs,
                                            (N, gh, gw, cur.shape[-1]), dtype,
                                            want_bits=save and self.fuse_gate and self.gate_bits)
                    if got is not None:
                        u, ds, y_out, bits = got
                        if dsc.stride[0] > 1:
                            ds.lo = (dsc.stride[0], tuple(y.shape))
                        units.append(u)
                        if save:
                            blocks.append(BlockRec(y, units, ds, y_out, gh * gw, si, bi == nb - 1, gate_bits=bits))
                        y, h, w = y_out, gh, gw
                        continue
                    # no two-source kernel for this shape: conv3 in the ordinary way
                    u = self._unit_fwd(conv3, main[-1][1], False, cur, cur_pro, (N, gh, gw, cur.shape[-1]), dtype)
                    units.append(u)
                    cur, cur_pro, gh, gw = u.c, u.st, u.desc.P, u.desc.Q
                if fused_tail:
                    u, y_out, bits = self._conv_bn_res_fwd(conv3, main[-1][1], cur, cur_pro, y,
                                                           (N, gh, gw, cur.shape[-1]), dtype,
                                                           want_bits=save and self.fuse_gate and self.gate_bits)
                    units.append(u)
                    last = u
                    if save:
                        blocks.append(BlockRec(y, units, ds, y_out, gh * gw, si, bi == nb - 1, gate_bits=bits))
                    y, h, w = y_out, gh, gw
                    continue
                last = units[-1]
                y_out = torch.empty_like(last.c)
                if blk.downsample is not None:
                    ds = self._unit_fwd(blk.downsample[0], blk.downsample[1], False, y, None, (N, h, w, cin), dtype)
                    kn.bn_act(last.c, last.st.scale, last.st.shift, y_out, ident=ds.c, id_scale=ds.st.scale,
                              id_shift=ds.st.shift, relu=True)
                else:
                    kn.bn_act(last.c, last.st.scale, last.st.shift, y_out, ident=y, relu=True)
                if save:
                    if (self._drop_c3 or self.fold_bn3) and len(units) == 3:
                        # Bottleneck: the 4x-wide conv3 output is 1/3 of the kept bytes.  With the folded bn3
                        # backward (_block_end_folded) it is never needed again; otherwise _block_bwd re-runs the
                        # 1x1 conv3 from the kept c2 with the kept statistics
                        last.c = None
                    rec_b = BlockRec(y, units, ds, y_out, gh * gw, si, bi == nb - 1)
                    if self._foldable(rec_b) and self._ds_foldable(ds):
                        ds.c = None  # the folded downsample backward never reads the branch output either
                    blocks.append(rec_b)
                y, h, w = y_out, gh, gw
            f = torch.empty(N, y.shape[-1], dtype=dtype, device=x.device)
            # the next stage's strided downsample branch reads y[:, ::2, ::2, :]: written by the pooling pass over y
            nxt = stages_list[si + 1][0] if si + 1 < len(stages_list) else None
            nds = getattr(nxt, "downsample", None)
            strided_next = None
            if (self.gap_stride_fused and nds is not None and isinstance(nds[0], nn.Conv2d) and nds[0].stride == (2, 2)
                    and nds[0].kernel_size == (1, 1) and self.fold_ds_strided and hasattr(nxt, "conv3")):
                strided_next = torch.empty(N, (h + 1) // 2, (w + 1) // 2, y.shape[-1], dtype=dtype, device=x.device)
                kn.gap_fwd_stride2(y, f, strided_next, N, h, w, y.shape[-1])
            else:
                kn.gap_fwd(y, f, N, h * w, y.shape[-1])
            feats.append(f)
        if not save:
            return EncPass(enc, N, H, W, None, None, None, None, [], feats, x_src=x, saved=False)
        return EncPass(enc, N, H, W, xin, stem, pooled, amax, blocks, feats)

    def _materialise(self, ps: EncPass, dtype: torch.dtype) -> EncPass:
        """re-run a features-only pass with activations kept; BatchNorm running statistics are not
        touched a second time"""
        if ps.saved:
            return ps
        keep = self.update_running
        self.update_running = False
        try:
            full = self.encoder_forward(ps.enc, ps.x_src, dtype, save=True)
        finally:
            self.update_running = keep
        return full

    def _plan_recompute(self, per_image_bytes: float, B: int, K: int, device, c3_fraction: float = 0.0,
                        shape_key: tuple = (), ctx_passes: int = 1) -> set:
        """which encoder passes run features-only in forward, and whether bottleneck conv3 outputs are dropped
        (engine.recompute = off | c3 | t1 | targets | auto).  Sets self._drop_c3 for the remaining passes.

        With more than one rank the decision is COLLECTIVE: a rank that recomputes a pass issues forward SyncBatchNorm
        exchanges where its peers issue backward ones (same message sizes, so a mismatch would pair silently), and
        'auto' depends on each rank's free memory.  Every rank therefore adopts the most conservative local plan
        (one all-reduce(MAX) of a plan code), once per (batch, model, image) shape -- the result is cached.  The cache
        key holds RANK-INVARIANT values only (batch, tile count, image / model shape, mode): whether the plan
        collective runs must never depend on a locally measured quantity, or one rank would issue it while its peers
        issue BatchNorm all-reduces (`per_image_bytes` comes from this rank's allocator and only feeds the LOCAL
        proposal that goes into the MAX)."""
        nosave, drop, pair = self._plan_local(per_image_bytes, B, K, device, c3_fraction, ctx_passes)
        if self._world() > 1:
            key = (B, K, tuple(shape_key), self._mode_override or self.recompute, self.fold_bn3)
            hit = self._plan_cache.get(key)
            if hit is None:
                # most conservative plan of all ranks: more features-only passes > drop conv3 > no side-by-side backward
                code = torch.tensor([4 * len(nosave) + 2 * int(drop) + int(not pair)], dtype=torch.int32, device=device)
                dist.all_reduce(code, op=dist.ReduceOp.MAX, group=self.group)
                self.collectives += 1
                c = int(code.item())
                hit = (frozenset([(), ("t1",), ("t0", "t1")][c // 4]), bool(c & 2), not (c & 1))
                self._plan_cache[key] = hit
            nosave, drop, pair = set(hit[0]), hit[1], hit[2]
        self._drop_c3 = drop
        self._pair_bwd = pair
        self.last_plan = ("recompute:" + ",".join(sorted(nosave)) if nosave else "keep-all") + (",drop-c3" if drop else "")
        return nosave

    def _plan_known(self, B: int, K: int, shape_key: tuple) -> bool:
        """the collective plan of this shape has been agreed (only ever true with more than one rank)"""
        key = (B, K, tuple(shape_key), self._mode_override or self.recompute, self.fold_bn3)
        return self._world() > 1 and key in self._plan_cache

    def plan_preview(self, B: int, K: int, shape_key: tuple, device, world: int) -> Optional[str]:
        """the LOCAL memory plan this rank would propose in a `world`-rank run of the shape it has just run (from the
        calibration of that shape and the memory free right now): what bench.py prints as `plan_at_8_ranks` on a
        one-GPU box.  None when the shape has not been calibrated."""
        cal = self._calib.get((B, K) + tuple(shape_key))
        if cal is None:
            return None
        self._world = lambda: world  # instance attribute shadows the method for the duration of the call
        try:
            nosave, drop, pair = self._plan_local(cal[0], B, K, device, cal[1], 2)
        finally:
            del self._world
        plan = ("recompute:" + ",".join(sorted(nosave)) if nosave else "keep-all") + (",drop-c3" if drop else "")
        if world > 1:
            # two sets of backward transients fit beside the RCCL reserve: the multi-stream schedule (side streams on their
            # own communicators); otherwise the two views of an encoder in lockstep on one stream
            if pair and not nosave and self.multirank_streams and self.allow_multistream:
                return plan + ",dual-stream" + ("+context-stream" if self.ctx_stream else "")
            return plan + ",views-lockstep" + ("" if pair else "(forward only)")
        return plan

    def _plan_local(self, per_image_bytes: float, B: int, K: int, device, c3_fraction: float, ctx_passes: int = 1):
        """this rank's own (features-only passes, drop conv3 outputs, two views' backward side by side) choice;
        ctx_passes: context passes whose activations are not allocated yet when this is called"""
        mode = self._mode_override or getattr(self, "recompute", "off")
        if self.fold_bn3 and c3_fraction > 0:  # conv3 outputs are never kept on the folded path
            per_image_bytes *= 1.0 - c3_fraction
            c3_fraction = 0.0
        free, _ = torch.cuda.mem_get_info(device)
        avail = free + torch.cuda.memory_reserved(device) - torch.cuda.memory_allocated(device)
        # head room: 6 GiB for the allocator's fragmentation; with more than one rank another 4 GiB for RCCL's channel
        # buffers and the IPC mappings of the peers, which are allocated outside torch's pool after this measurement
        # (prepare_multirank measured what the communicators took at creation -- already missing from `free`; the reserve
        # covers what RCCL adds on first use: at least 4 GiB, or TWICE what creation took -- with real peers the channel
        # buffers of four communicators are unmeasured here (one rank: 1.07 GiB), and a plan that is one allocation away from
        # the card's limit dies inside RCCL, where nothing can fall back; the cost of being wrong the other way is the
        # lockstep schedule instead of side streams, a few per cent)
        budget = avail - (6 << 30) - (max(4 << 30, 2 * self.comm_bytes) if self._world() > 1 else 0)
        # calibration (ResNet-50, 256 tile pairs, bf16): kept activations 180 GiB, measured peak 240.8 GiB with
        # 28 GiB of weights/optimizer -> backward transients (gradient tensors, re-normalised operands, the
        # recomputed conv3) are about a quarter of one full target pass
        one_target = per_image_bytes * B * K
        transients = 0.26 * one_target
        ctx = per_image_bytes * B * ctx_passes  # the second context pass (both, when the plan precedes the first)
        forced = {"off": (set(), False), "c3": (set(), c3_fraction > 0), "t1": ({"t1"}, False),
                  "targets": ({"t0", "t1"}, False)}.get(mode)
        if forced is not None:
            # MSFWSI_RECOMPUTE=<mode> fixes WHAT is kept; whether two sets of backward transients (and with them the
            # multi-stream schedule's extra allocator pools) fit beside it is still a question of memory
            kept = 2 - len(forced[0])
            slim = 1.0 - c3_fraction if forced[1] else 1.0
            return forced[0], forced[1], (ctx + kept * one_target) * slim + 2 * transients < budget
        if ctx + 2 * one_target + transients < budget:
            # lockstep backward of the two target passes holds two sets of transients at once
            return set(), False, ctx + 2 * one_target + 2 * transients < budget
        slim, drop = 1.0, False
        if c3_fraction > 0:
            slim = 1.0 - c3_fraction
            drop = True
            if ctx * slim + 2 * one_target * slim + transients < budget:
                return set(), drop, False
        if ctx * slim + one_target * slim + transients < budget:
            return {"t1"}, drop, False
        return {"t0", "t1"}, drop, False

    def encoder_backward(self, ps: EncPass, dfeats: Sequence[Optional[torch.Tensor]], grads: GradStore,
                         dtype: torch.dtype, dmaps: Optional[Sequence[Optional[torch.Tensor]]] = None,
                         dstem: Optional[torch.Tensor] = None):
        """dfeats: gradients of the four pooled features.  dmaps / dstem (U-Net skip connections, unet_engine): dense
        gradients of the four stage OUTPUT maps [N,H,W,C] and of the stem activation relu(bn1(conv1)) [N,H/2,W/2,64]."""
        dy, pre = None, None
        for i in range(len(ps.blocks) - 1, -1, -1):
            rec = ps.blocks[i]
            gapg = dfeats[rec.stage] if rec.stage_end else None
            if dmaps is not None and rec.stage_end and dmaps[rec.stage] is not None:
                dm = dmaps[rec.stage]
                if dy is None:
                    dy = dm
                else:  # dy + dm through the BatchNorm-apply kernel with the unit map (one rounding)
                    one, zero = self._unit_gate(dm.shape[-1], dm.device)
                    tot = torch.empty_like(dy)
                    kn.bn_act(dy, one, zero, tot, ident=dm, relu=False)
                    dy = tot
            if dy is None and gapg is None:
                raise RuntimeError("encoder_backward: no gradient reaches the last block")
            gate = None
            if i > 0 and self.fuse_gate and dmaps is None:
                pr = ps.blocks[i - 1]
                if self._foldable(pr) and (pr.ds is None or self._ds_foldable(pr.ds)):
                    # the producer of pr's output gradient (this block's first conv) applies pr's closing ReLU gate,
                    # adds pr's pooled-feature gradient and reduces sum(g) in its own epilogue
                    gate = (pr.y_out, dfeats[pr.stage] if pr.stage_end else None, pr.HW, pr.gate_bits)
            dy, pre = self._block_bwd(rec, dy, gapg, grads, dtype, pre=pre, gate=gate)
            rec.units = []  # release activations
            rec.ds = None
        self._stem_bwd(ps, dy, grads, dtype, dstem)

    def _stem_bwd(self, ps: EncPass, dy: torch.Tensor, grads: GradStore, dtype: torch.dtype,
                  dstem: Optional[torch.Tensor] = None):
        """stem: maxpool + relu + bn1 backward, then conv1's weight gradient (resnet.py:234-237 backwards); dy = gradient
        of the max-pool output [N,P,Q,64]"""
        st, u = ps.stem.st, ps.stem
        H0, W0 = u.desc.P, u.desc.Q
        sums = kn.new_stats(64, 2, u.c.device)
        g0 = torch.empty_like(u.c)
        kn.stem_pool_bwd(dy, ps.amax, u.c, st.scale, st.shift, g0, sums, ps.N, H0, W0, 64, dact=dstem)
        k = self._bn_bwd_coeffs(sums, 2, 1, u.bn, st, grads)
        if u.s2d and self.stem_fuse_bnbwd:
            # dc0 = k1*g + k2*c0 + k3 formed inside the weight-gradient kernel's staging: the 6.6 GB gradient is
            # neither rewritten nor re-read (-13 GB per target pass)
            dw2 = kn.zeros((u.desc.K, 4, 4, 16), torch.float32, g0.device)
            if kn.stem_wgrad_bnbwd(u.desc, u.x, g0, u.c, (k[0], k[1], k[2]), dw2):
                kn.stem_s2d_wfold(dw2, grads.get(u.op.weight))
                return
        kn.bn_bwd_apply(g0, u.c, k[0], k[1], k[2], g0)
        self._unit_wgrad(u, g0, grads, dtype)

    def _foldable(self, rec: BlockRec) -> bool:
        """Bottleneck block whose closing 1x1 conv (conv3) + BatchNorm backward can be folded into weights"""
        if not self.fold_bn3 or len(rec.units) != 3:
            return False
        d = rec.units[-1].desc
        return d.R == 1 and d.S == 1 and d.stride == 1 and rec.units[-1].x_pro is not None \
            and getattr(rec.units[-1].op, "bias", None) is None

    def _ds_foldable(self, ds: Optional[Unit]) -> bool:
        """downsample branch (1x1 conv + BatchNorm on the block input) whose BatchNorm backward folds into weights the
        same way: stride 1 only (layer1.0) -- a strided branch would need the Gram matrix of a strided gather"""
        if ds is None or not self.fold_ds:
            return False
        d = ds.desc
        bk = 16 if d.dtype == 0 else 32
        return (d.R == 1 and d.S == 1 and d.stride == 1 and ds.x_pro is None and getattr(ds.op, "bias", None) is None
                and d.C % bk == 0 and d.K % bk == 0)

    def _ds_folded_bwd(self, rec: BlockRec, g: torch.Tensor, sums: torch.Tensor, ns: int, grads: GradStore, dtype):
        """input gradient of the downsample branch y_d = bn_d(W_d x) from the gated block-output gradient g, without
        the branch output c_d (same algebra as _block_end_folded, the operand is the block input x itself):
        returns d(x) of that branch = g (k1 o W_d) + x (W_d^T diag(k2) W_d) + W_d^T k3"""
        u = rec.ds
        d = u.desc
        K, Ci = d.K, d.C
        dev = g.device
        x = u.x
        Wd = WeightStore.physical(u.op.weight).view(K, 1, 1, Ci)
        Md = kn.zeros((K, 1, 1, Ci), torch.float32, dev)
        kn.conv_wgrad(d, x, g, Md)
        if u.gram is not None:  # kept by the two-source forward
            Ax, sx = u.gram
        else:
            Ax = kn.zeros((Ci, 1, 1, Ci), torch.float32, dev)
            kn.gram(kn.conv_desc(dtype, d.N, d.H, d.W, Ci, Ci, 1, 1, 1, 0), x, Ax)
            sx = kn.zeros((Ci,), torch.float64, dev)
            kn.colsum(x, sx)
        packed = torch.empty(ns * K, dtype=torch.float64, device=dev)
        kn.shard_sum(sums, packed)  # slot 0 = sum(g) (local)
        sd = kn.zeros((1, 2, K), torch.float64, dev)
        kn.shard_sum(packed[:K], sd[0, 0])  # one "shard": a device copy without a torch operator
        kn.fold_dots(Wd, Md, sd[0, 1])
        kd = self._bn_bwd_coeffs(sd, 2, 1, u.bn, u.st, grads)
        dlin = kn.conv_desc(torch.float32, K, 1, 1, Ci, Ci, 1, 1, 1, 0)
        WA = torch.empty(K, 1, 1, Ci, dtype=torch.float32, device=dev)
        kn.conv_fwd(dlin, Wd, Ax, WA)
        # the two-source weights [k1 o W ; W^T diag(k2) W] are built in place: both parts are row blocks of one matrix
        wcat32 = kn.zeros((K + Ci, Ci), torch.float32, dev)
        Wk1, Gd = wcat32[:K].view(K, 1, 1, Ci), wcat32[K:].view(Ci, 1, 1, Ci)
        Wk2 = torch.empty_like(WA)
        bvec = kn.zeros((Ci,), torch.float32, dev)
        kn.fold_weights(Wd, Md, WA, kd[0], kd[1], kd[2], sx, grads.get(u.op.weight), Wk1, Wk2, bvec)
        kn.conv_wgrad(dlin, Wd, Wk2, Gd)
        wcat = wcat32 if dtype == torch.float32 else kn.cast_lowp(wcat32, torch.empty_like(wcat32, dtype=dtype))
        dx = torch.empty_like(x)
        if self.fuse_two_source and kn.conv_dgrad2(d, g, wcat, dx, x, bias=bvec):
            return dx
        t = torch.empty_like(x)
        kn.conv_fwd(kn.conv_desc(dtype, d.N, d.H, d.W, Ci, Ci, 1, 1, 1, 0), x, wcat[K:].view(Ci, 1, 1, Ci), t, bias=bvec)
        kn.conv_dgrad(d, g, wcat[:K].view(K, 1, 1, Ci), dx, resid=t)
        return dx

    def _block_end_folded(self, rec: BlockRec, dy, gapg, grads: GradStore, dtype, pre=None):
        """Backward through y = relu(bn3(conv3(a2)) + identity) WITHOUT the 4x-wide conv3 output c3 = W a2:
        every c3-dependent term of the BatchNorm backward dc3 = k1*g + k2*c3 + k3 is folded into [K][C] / [C][C]
        matrices (msfwsi_fold_dots / msfwsi_fold_weights), so c3 is neither kept, re-made nor re-read:
            sum g*c3 = rowdot(W, M),  M = g^T a2              (one weight-gradient launch on g)
            dW      += k1 o M + k2 o (W A) + k3 (x) sum(a2),  A = a2^T a2
            da2      = [ g (k1 o W) + a2 (W^T diag(k2) W) + W^T k3 ] * relu'(bn2(c2))
        Returns (g, da2_gated, sums2 of the gated da2 for bn2's backward, the downsample coefficients or None)."""
        last, prev = rec.units[-1], rec.units[-2]
        d = last.desc
        K, Cw = d.K, d.C
        dev = rec.y_out.device
        ds_fold = self._ds_foldable(rec.ds)
        if pre is not None:
            # dy is already the gated gradient (ReLU gate + pooled-feature gradient applied by its producer);
            # pre = [nshard][2][K] with slot 0 = sum(g); slot 1 is overwritten by fold_dots below
            g, sums, ns = dy, pre, 2
            kn.zero_slot(sums, 1)
        else:
            g, ns = torch.empty_like(rec.y_out), 3
            sums = kn.new_stats(K, 3, dev)
            kn.block_end_bwd(dy, rec.y_out, gapg, 1.0 / rec.HW, None,
                             rec.ds.c if rec.ds is not None and not ds_fold else None, g, sums, rec.HW)
        W = WeightStore.physical(last.op.weight).view(K, 1, 1, Cw)
        Mm = kn.zeros((K, 1, 1, Cw), torch.float32, dev)
        # Round 6: the two-source launch below takes conv2's RAW output as its second source and forms a2 = relu(bn2(c2)) on
        # the fragments it reads from LDS (msfwsi_conv_dgrad2_pro: the arithmetic of bn_act, bit for bit) -- the same tensor
        # it reads for the gate.  a2 is then needed only by the M = g^T a2 launch: where that launch normalises its operand
        # in its own staging (64 channels) a2 is never stored at all; elsewhere the bn_act pass stays (the DMA-staged
        # weight-gradient kernel wants a materialised operand) but its output is read once instead of twice
        src2_in_launch = (self.dgrad2_pro and Cw <= self.dgrad2_pro_max_c and self.fuse_two_source
                          and dtype != torch.float32 and prev.c is last.x and last.x_pro is prev.st)
        a2 = None
        fused_a2 = False
        if last.gram is not None:  # Gram matrix and column sums of a2 kept by the fused forward
            A, sa = last.gram
            # 64 channels (layer1, HBM-bound): M = g^T a2 with bn2 + ReLU applied in the weight-gradient kernel's register
            # staging (which also writes a2 where the two-source launch still wants it) -- 1.41 (+ the write) against
            # 0.68 + 1.57 ms of bn_act + the DMA-staged launch; wider layers lose with register staging
            # (profiles/r05_kbench_mwgrad.txt)
            if self.fuse_a2_wgrad and Cw <= self.fuse_a2_wgrad_max_c and dtype != torch.float32:
                if src2_in_launch:   # nobody reads a2: the same register-staged launch without the by-product
                    kn.conv_wgrad(d, last.x, g, Mm, pro=(last.x_pro.scale, last.x_pro.shift))
                    fused_a2 = True
                else:
                    a2 = torch.empty_like(last.x)
                    fused_a2 = kn.conv_wgrad_act(d, last.x, g, Mm, (last.x_pro.scale, last.x_pro.shift), a2)
                    if not fused_a2:
                        a2 = None
            if not fused_a2:
                a2 = torch.empty_like(last.x)
                kn.bn_act(last.x, last.x_pro.scale, last.x_pro.shift, a2, relu=True)
        else:  # relu(bn2(c2)) and its column sums in one pass
            A = None
            sa = kn.zeros((Cw,), torch.float64, dev)
            a2 = torch.empty_like(last.x)
            kn.bn_act_sum(last.x, last.x_pro.scale, last.x_pro.shift, a2, sa)
        if not fused_a2:
            kn.conv_wgrad(d, a2, g, Mm)
        dsq = kn.conv_desc(dtype, d.N, d.P, d.Q, Cw, Cw, 1, 1, 1, 0)
        if A is None:
            A = kn.zeros((Cw, 1, 1, Cw), torch.float32, dev)
            kn.gram(dsq, a2, A)
        kn.fold_dots(W, Mm, sums[0, 1])  # slot 1 of shard 0; the other shards of that slot stay zero
        k = self._bn_bwd_coeffs(sums, ns, 1, last.bn, last.st, grads)
        kd = self._bn_bwd_coeffs(sums, 3, 2, rec.ds.bn, rec.ds.st, grads) if rec.ds is not None and not ds_fold else None
        resid_ds = self._ds_folded_bwd(rec, g, sums, ns, grads, dtype) if ds_fold else None
        # small fp32 matrices on the exact-fp32 MFMA path: WA = W A, G = W^T diag(k2) W
        dlin = kn.conv_desc(torch.float32, K, 1, 1, Cw, Cw, 1, 1, 1, 0)
        WA = torch.empty(K, 1, 1, Cw, dtype=torch.float32, device=dev)
        kn.conv_fwd(dlin, W, A, WA)  # A is symmetric: [out][in] == [in][out]
        # [k1 o W ; G] are the two row blocks of ONE matrix (the two-source launch's weights), built in place
        wcat32 = kn.zeros((K + Cw, Cw), torch.float32, dev)
        Wk1, G = wcat32[:K].view(K, 1, 1, Cw), wcat32[K:].view(Cw, 1, 1, Cw)
        Wk2 = torch.empty_like(WA)
        bvec = kn.zeros((Cw,), torch.float32, dev)
        kn.fold_weights(W, Mm, WA, k[0], k[1], k[2], sa, grads.get(last.op.weight), Wk1, Wk2, bvec)
        kn.conv_wgrad(dlin, W, Wk2, G)  # G[i][j] = sum_k k2[k] W[k][i] W[k][j]
        s2 = kn.new_stats(Cw, 2, dev)
        da = torch.empty_like(last.x)
        gate = (prev.c, prev.st.scale, prev.st.shift)
        # one launch: da2 = gate([g | a2] . [k1 o W ; G] + W^T k3), the k range of a2 follows the one of g
        wcat = wcat32 if dtype == torch.float32 else kn.cast_lowp(wcat32, torch.empty_like(wcat32, dtype=dtype))
        if src2_in_launch and kn.conv_dgrad2(d, g, wcat, da, last.x, bias=bvec, mask=gate, sums=s2,
                                             src2_pro=(last.x_pro.scale, last.x_pro.shift)):
            return g, da, s2, kd, resid_ds
        if a2 is None:  # (the launch declined the shape after the weight gradient had skipped the by-product)
            a2 = torch.empty_like(last.x)
            kn.bn_act(last.x, last.x_pro.scale, last.x_pro.shift, a2, relu=True)
        if self.fuse_two_source and kn.conv_dgrad2(d, g, wcat, da, a2, bias=bvec, mask=gate, sums=s2):
            return g, da, s2, kd, resid_ds
        # shapes without a two-source kernel: the a2 term as a separate w -> w conv, added as the residual
        Wc, Gc = wcat[:K].view(K, 1, 1, Cw), wcat[K:].view(Cw, 1, 1, Cw)
        t = torch.empty_like(a2)
        kn.conv_fwd(dsq, a2, Gc, t, bias=bvec)
        kn.conv_dgrad(d, g, Wc, da, resid=t, mask=gate, sums=s2)
        return g, da, s2, kd, resid_ds

    def _block_bwd(self, rec: BlockRec, dy, gapg, grads: GradStore, dtype, pre=None, gate=None):
        """returns (gradient w.r.t. the block input, its fused-gate sums or None -- see encoder_backward)"""
        if self._foldable(rec):
            g, da, s2, kd, resid_ds = self._block_end_folded(rec, dy, gapg, grads, dtype, pre=pre)
            resid, rstride = g, 1
            if resid_ds is not None:
                resid = resid_ds  # the skip connection IS the (folded) downsample branch
                if rec.ds.lo is not None:  # strided branch: its input gradient lives on the subsampled pixels
                    if self.lores_resid:
                        rstride = rec.ds.lo[0]  # added on that sub-grid by conv1's input-gradient epilogue
                    else:
                        resid = torch.empty(rec.ds.lo[1], dtype=dtype, device=resid_ds.device)
                        kn.pixel_stride(resid_ds, resid, rec.ds.lo[0], expand=True)
            elif rec.ds is not None:
                kn.bn_bwd_apply(g, rec.ds.c, kd[0], kd[1], kd[2], g)  # g becomes d(downsample conv output)
                self._unit_wgrad(rec.ds, g, grads, dtype)
                resid = self._unit_dgrad(rec.ds, g, dtype)
            prev = rec.units[1]
            kp = self._bn_bwd_coeffs(s2, 2, 1, prev.bn, prev.st, grads)
            top_bn = None
            if self._img3_dgrad_weights(prev, rec.units[0], dtype) is not None:
                top_bn = (prev.c, kp)  # bn2's backward is formed inside conv2's input-gradient launch (image-stationary kernel)
            else:
                kn.bn_bwd_apply(da, prev.c, kp[0], kp[1], kp[2], da)
            return self._block_bwd_tail(rec, da, 1, resid, grads, dtype, gate=gate, resid_stride=rstride, top_bn=top_bn)
        if pre is not None:
            raise RuntimeError("a pre-gated gradient reached a block that is not on the folded path")
        last = rec.units[-1]
        last_xmat = None
        if last.c is None:  # dropped bottleneck conv3 output: re-run the 1x1 conv from the kept c2 + statistics
            d = last.desc
            last.c = torch.empty(d.N, d.P, d.Q, d.K, dtype=dtype, device=rec.y_out.device)
            pro = (last.x_pro.scale, last.x_pro.shift) if last.x_pro is not None else None
            xin = last.x
            if pro is not None:
                # the same normalised operand the forward used; shared with this conv's weight gradient below
                last_xmat = self._normalised_operand(last)
                xin, pro = last_xmat, None
            kn.conv_fwd(d, xin, self.weights.get(last.op.weight, dtype), last.c, pro=pro)
        Cn = last.c.shape[-1]
        dev = last.c.device
        g = torch.empty_like(rec.y_out)
        sums = kn.new_stats(Cn, 3, dev)
        kn.block_end_bwd(dy, rec.y_out, gapg, 1.0 / rec.HW, last.c, rec.ds.c if rec.ds is not None else None, g, sums,
                         rec.HW)
        k = self._bn_bwd_coeffs(sums, 3, 1, last.bn, last.st, grads)
        dc = torch.empty_like(g)
        kn.bn_bwd_apply(g, last.c, k[0], k[1], k[2], dc)
        resid = g
        if rec.ds is not None:
            kd = self._bn_bwd_coeffs(sums, 3, 2, rec.ds.bn, rec.ds.st, grads)
            kn.bn_bwd_apply(g, rec.ds.c, kd[0], kd[1], kd[2], g)  # g becomes d(downsample conv output)
            self._unit_wgrad(rec.ds, g, grads, dtype)
            resid = self._unit_dgrad(rec.ds, g, dtype)
        return self._block_bwd_tail(rec, dc, len(rec.units) - 1, resid, grads, dtype, last_xmat, gate=gate)

    def _unit_gate(self, Cn: int, dev):
        key = (Cn, str(dev))
        if key not in self._gate_vecs:
            self._gate_vecs[key] = (torch.ones(Cn, dtype=torch.float32, device=dev),
                                    torch.zeros(Cn, dtype=torch.float32, device=dev))
            torch.cuda.current_stream(dev).synchronize()  # first use only: both streams may read them from now on
        return self._gate_vecs[key]

    def _panel_dgrad_ok(self, first: Unit, resid, gate) -> bool:
        """conv1's input gradient on the panel kernel: 16-bit 1x1 / stride 1 with a panel-sized k range, the gate (if any)
        as bits; the stem-side conv of a BasicBlock (3x3) never qualifies"""
        if not self.panel_dgrad or first.x_pro is not None or getattr(first.op, "bias", None) is not None or first.s2d:
            return False
        if gate is not None and gate[3] is None:  # the previous block's gate only as its activation: gather kernel
            return False
        return kn.panel_supported(first.desc, True)

    def _img3_fills(self, d, dev) -> bool:
        """an image-stationary launch has one workgroup per band of an image: it pays only when the bands fill the chip at
        least once (ResNet-18 at 8 tile pairs has 128 images per view: 128 workgroups of 55 us each on 256 CUs were 10 %
        of the step slower than the gather kernel's 392 small tiles, profiles/r05_ab_small_batches.txt)"""
        rows = d.P if d.stride == 2 else d.H  # the staged tensor: the gradient for the strided layer
        bands, per_cu = {14: (1, 1), 28: (4, 2), 56: (14, 3)}.get(rows, (0, 1))
        ncu = self._ncu.get(dev)
        if ncu is None:
            ncu = self._ncu[dev] = torch.cuda.get_device_properties(dev).multi_processor_count
        return d.N * bands >= self.img3x3_min_fill * ncu * per_cu

    def _img3_bwd_chunks(self, d, a1_like: Optional[torch.Tensor], dev) -> int:
        """image chunks of the gradient-launch + weight-gradient pair: 1 unless the by-product activation exceeds
        img3x3_chunk_bytes (1 GiB) and every chunk still fills the chip twice"""
        if a1_like is None:
            return 1
        nbytes = a1_like.numel() * a1_like.element_size()
        n = 1
        while (nbytes // n > self.img3x3_chunk_bytes and n < 8 and d.N % (2 * n) == 0
               and self._img3_fills(types.SimpleNamespace(N=d.N // (2 * n) // 2, H=d.H, P=d.P, stride=d.stride), dev)):
            n *= 2
        return n

    def _img3_dgrad_weights(self, u: Unit, prev: Unit, dtype, with_bn: bool = True) -> Optional[torch.Tensor]:
        """the packed filter if u's input gradient runs on the image-stationary kernel: a served 3x3 geometry whose operand
        is prev's raw output under prev's BatchNorm + ReLU (conv2 of a Bottleneck of layer2 / layer3; of layer1 only when the
        launch also forms the BatchNorm backward, with_bn), else None"""
        if u.x_pro is None or u.x is not prev.c or prev.st is None or u.s2d:
            return None
        if not (kn.img3x3_supported(u.desc) or kn.img3x3_s2_dgrad_supported(u.desc)):
            return None
        if not self._img3_fills(u.desc, prev.c.device):
            return None
        if u.desc.C == 64 and not with_bn:
            return None
        return self._img3_weights(u.op, self.weights.get(u.op.weight, dtype), dtype, dgrad=True)

    def _block_bwd_tail(self, rec: BlockRec, cur, top: int, resid, grads: GradStore, dtype, last_xmat=None,
                        gate=None, resid_stride: int = 1, top_bn=None):
        """units[top] .. units[0]: weight gradient, input gradient with the producer's ReLU gate + BatchNorm sums
        fused into its epilogue, BatchNorm backward; the first unit adds the identity-path gradient `resid`.
        top_bn = (c, (k1, k2, k3)): `cur` is still the gradient w.r.t. units[top]'s BatchNorm OUTPUT and the caller has
        checked (_img3_dgrad_weights) that units[top]'s input-gradient launch forms the BatchNorm backward itself."""
        dev = cur.device
        fused = None
        for i in range(top, 0, -1):
            u, prev = rec.units[i], rec.units[i - 1]
            s2 = kn.new_stats(prev.c.shape[-1], 2, dev)
            bn_here = top_bn if i == top else None
            wimg = self._img3_dgrad_weights(u, prev, dtype, with_bn=bn_here is not None)
            if wimg is not None:
                # ONE launch: dc = bn backward of `cur` while the band is staged (written for the weight gradient: in place
                # where a workgroup owns the whole image), da = gate(conv^T(dc)) + bn1's sums, and a1 = relu(bn1(c1)) --
                # the weight gradient's operand -- stored from the gate's own arithmetic
                d = u.desc
                da = torch.empty(d.N, d.H, d.W, d.C, dtype=dtype, device=dev)
                strided = d.stride == 2
                want_a1 = d.C != 64  # (64 channels: the output-stationary weight-gradient kernel normalises c1 in its own staging)
                inplace = (d.P if strided else d.H) == 14  # a workgroup owns the whole (gradient) image: bands read their neighbours' halo rows
                a1 = dc = bnb = None
                if bn_here is not None:
                    bnb = (bn_here[0], bn_here[1][0], bn_here[1][1], bn_here[1][2])
                chunked = want_a1 and self._img3_bwd_chunks(d, prev.c, dev) > 1
                if not chunked:
                    a1 = torch.empty_like(prev.c) if want_a1 else None
                    dc = cur
                    if bnb is not None and not inplace:
                        dc = torch.empty_like(cur)
                elif inplace:
                    dc = cur
                launch = kn.img3x3_s2_dgrad if strided else kn.img3x3_dgrad
                nchunk = self._img3_bwd_chunks(d, prev.c, dev) if chunked else 1
                if nchunk > 1:
                    # the by-product a1 is as large as da (3.3 GB per view at 56 x 56 x 128) and both live until the weight
                    # gradient has run: with two views in flight that raised the allocator's pools by 9.5 GiB, to the card's
                    # last GiB.  In chunks of images the launch + weight-gradient pair re-uses ONE chunk-sized a1 / dc
                    # (the sums and the weight gradient accumulate; a chunk still fills the chip several times).
                    nc = d.N // nchunk
                    dch = kn.conv_desc(dtype, nc, d.H, d.W, d.C, d.K, d.R, d.S, d.stride, d.pad)
                    a1c = torch.empty((nc,) + tuple(prev.c.shape[1:]), dtype=dtype, device=dev)
                    dcc = torch.empty((nc,) + tuple(cur.shape[1:]), dtype=dtype, device=dev) if bnb is not None and dc is not cur else None
                    dw = grads.get(u.op.weight)
                    for ci in range(nchunk):
                        sl = slice(ci * nc, (ci + 1) * nc)
                        dco = (dcc if dcc is not None else cur[sl]) if bnb is not None else None
                        if not launch(dch, cur[sl], wimg, da[sl], bnbwd=(bnb[0][sl],) + tuple(bnb[1:]) if bnb is not None else None,
                                      dc_out=dco, mask=(prev.c[sl], prev.st.scale, prev.st.shift), sums=s2, act_out=a1c):
                            raise _lib.MsfwsiHipError("the image-stationary gradient refused a geometry its `supported` accepted")
                        kn.conv_wgrad(dch, a1c, dco if dco is not None else cur[sl], dw)
                    del a1c, dcc
                else:
                    if not launch(d, cur, wimg, da, bnbwd=bnb, dc_out=dc if bnb is not None else None,
                                  mask=(prev.c, prev.st.scale, prev.st.shift), sums=s2, act_out=a1):
                        raise _lib.MsfwsiHipError("the image-stationary gradient refused a geometry its `supported` accepted")
                    self._unit_wgrad(u, dc, grads, dtype, x_mat=a1)
                del a1, dc
            else:
                if bn_here is not None:
                    raise RuntimeError("top_bn without the image-stationary kernel: the caller applies the BatchNorm backward")
                self._unit_wgrad(u, cur, grads, dtype, x_mat=last_xmat)
                # ReLU gate of prev and its BatchNorm-backward sums are fused into the dgrad epilogue
                da = self._unit_dgrad(u, cur, dtype, mask=(prev.c, prev.st.scale, prev.st.shift), sums=s2)
            last_xmat = None
            kp = self._bn_bwd_coeffs(s2, 2, 1, prev.bn, prev.st, grads)
            if i == 1 and self._panel_dgrad_ok(rec.units[0], resid, gate):
                fused = (da, prev.c, kp)  # bn1's backward is formed inside conv1's input-gradient launch below
            else:
                kn.bn_bwd_apply(da, prev.c, kp[0], kp[1], kp[2], da)
            cur = da
        first = rec.units[0]
        if fused is not None or (top == 0 and self._panel_dgrad_ok(first, resid, gate)):
            # panel kernel: dc1 = k1*g + k2*c1 + k3 while the operand panel is staged, written back IN PLACE for the weight
            # gradient (a workgroup reads exactly the rows it rewrites); epilogue as msfwsi_conv_dgrad's
            d1 = first.desc
            wpk = self._panel_weights(first.op, self.weights.get(first.op.weight, dtype), dtype, dgrad=True)
            dx = torch.empty(d1.N, d1.H, d1.W, d1.C, dtype=dtype, device=dev)
            y_prev, gapg_prev, hw_prev, bits_prev = gate if gate is not None else (None, None, 1, None)
            sg = kn.new_stats(d1.C, 2, dev) if bits_prev is not None else None
            bn = (fused[1], fused[2][0], fused[2][1], fused[2][2]) if fused is not None else None
            if kn.panel_dgrad(d1, cur, wpk, dx, bnbwd=bn, dc_out=cur if bn is not None else None, resid=resid,
                              resid_stride=resid_stride, gapg=gapg_prev, gap_scale=1.0 / hw_prev, mask_bits=bits_prev, sums=sg):
                self._unit_wgrad(first, cur, grads, dtype)
                return dx, sg
            if fused is not None:  # (not reached: _panel_dgrad_ok asked the library) -- the stand-alone pass after all
                kn.bn_bwd_apply(cur, fused[1], fused[2][0], fused[2][1], fused[2][2], cur)
        self._unit_wgrad(first, cur, grads, dtype)
        if gate is None:
            return self._unit_dgrad(first, cur, dtype, resid=resid, resid_stride=resid_stride), None
        y_prev, gapg_prev, hw_prev, bits_prev = gate
        Cn = y_prev.shape[-1]
        sg = kn.new_stats(Cn, 2, dev)
        if bits_prev is not None:  # 1/16 of the bytes of y_prev
            dx = self._unit_dgrad(first, cur, dtype, resid=resid, gapg=gapg_prev, gap_scale=1.0 / hw_prev,
                                  mask_bits=bits_prev, sums=sg, resid_stride=resid_stride)
        else:
            one, zero = self._unit_gate(Cn, dev)
            dx = self._unit_dgrad(first, cur, dtype, resid=resid, gapg=gapg_prev, gap_scale=1.0 / hw_prev,
                                  mask=(y_prev, one, zero), sums=sg, resid_stride=resid_stride)
        return dx, sg

    # ---- MLP heads -----------------------------------------------------------------------------
    @staticmethod
    def _parse_chain(seq: nn.Sequential):
        plan, cur = [], None
        for m in seq:
            if isinstance(m, nn.Linear):
                if cur is not None:
                    plan.append(cur)
                cur = [m, None, False]
            elif isinstance(m, nn.modules.batchnorm._BatchNorm):
                cur[1] = m
            elif isinstance(m, nn.ReLU):
                cur[2] = True
            else:
                raise NotImplementedError(f"unsupported head layer {type(m).__name__}")
        plan.append(cur)
        return plan

    def chain_forward(self, seq: nn.Sequential, x: torch.Tensor, dtype) -> ChainRec:
        rows = x.shape[0]
        cur, cur_pro = x, None
        units: List[Unit] = []
        out = None
        for lin, bn, relu in self._parse_chain(seq):
            u = self._unit_fwd(lin, bn, relu, cur, cur_pro, (rows, 1, 1, lin.in_features), dtype)
            units.append(u)
            if bn is not None and not relu:
                out = torch.empty_like(u.c)
                kn.bn_act(u.c, u.st.scale, u.st.shift, out, relu=False)
                cur, cur_pro = out, None
            elif bn is not None:
                cur, cur_pro, out = u.c, u.st, None
            else:
                cur, cur_pro, out = u.c, None, u.c
        if out is None:
            raise NotImplementedError("a head must end in Linear or BatchNorm (no trailing ReLU)")
        return ChainRec(units, out.view(rows, -1))

    def chain_forward_pair(self, seq: nn.Sequential, xs: Sequence[torch.Tensor], dtype) -> List[ChainRec]:
        """chain_forward for the two views of one head at once: every Linear runs ONCE on the stacked rows of both
        views (the 18432-wide fuser layers have 256 rows per view and are bound by reading their 680 MB weight matrix:
        once instead of twice), while every BatchNorm keeps its per-view batch -- statistics, running-statistics
        updates (view 0 then view 1, backbone.py:161-186) and the normalisation are per view on the row halves."""
        rows = [x.shape[0] for x in xs]
        tot = sum(rows)
        offs = [0, rows[0]]
        dev = xs[0].device
        Cin0 = xs[0].shape[-1]
        cur = torch.empty(tot, Cin0, dtype=dtype, device=dev)
        for v, x in enumerate(xs):
            kn.copy2d(x.reshape(rows[v], Cin0), 0, Cin0, cur, offs[v] * Cin0, Cin0, rows[v], Cin0)
        units: List[List[Unit]] = [[], []]
        pros: List[Optional[BNState]] = [None, None]
        raw = cur  # per view: the raw operand of the current layer (rows of `raw`), normalised into `cur` when pro
        outs: List[Optional[torch.Tensor]] = [None, None]
        for lin, bn, relu in self._parse_chain(seq):
            Cin, K = lin.in_features, lin.out_features
            if pros[0] is not None:  # BatchNorm + ReLU of the previous layer, per view, into the stacked operand
                cur = torch.empty(tot, Cin, dtype=dtype, device=dev)
                for v in range(2):
                    kn.bn_act(raw[offs[v]:offs[v] + rows[v]], pros[v].scale, pros[v].shift,
                              cur[offs[v]:offs[v] + rows[v]], relu=True)
            d = kn.conv_desc(dtype, tot, 1, 1, Cin, K, 1, 1, 1, 0)
            w = self.weights.get(lin.weight, dtype)
            c = torch.empty(tot, 1, 1, K, dtype=dtype, device=dev)
            bias = getattr(lin, "bias", None)
            kn.conv_fwd(d, cur.view(tot, 1, 1, Cin), w, c, bias=bias.data if bias is not None else None)
            nxt_pro: List[Optional[BNState]] = [None, None]
            sts2: List[Optional[BNState]] = [None, None]
            if bn is not None:
                stats2 = []
                for v in range(2):
                    stats = kn.new_stats(K, 2, dev)
                    kn.colstats(c[offs[v]:offs[v] + rows[v]], stats)  # fp64 column statistics (see _unit_fwd: the heads' cancellation problem)
                    stats2.append(stats)
                sts2 = self._bn_finalize_views(stats2, rows, bn)
            for v in range(2):
                cv = c[offs[v]:offs[v] + rows[v]]
                xv = raw[offs[v]:offs[v] + rows[v]].view(rows[v], 1, 1, Cin)
                dv = kn.conv_desc(dtype, rows[v], 1, 1, Cin, K, 1, 1, 1, 0)
                u = Unit(lin, bn, relu, dv, xv, pros[v], cv)
                if bn is not None:
                    u.st = sts2[v]
                units[v].append(u)
                if bn is not None and not relu:
                    out = torch.empty(rows[v], K, dtype=dtype, device=dev)
                    kn.bn_act(cv, u.st.scale, u.st.shift, out, relu=False)
                    outs[v] = out
                elif bn is not None:
                    nxt_pro[v], outs[v] = u.st, None
                else:
                    outs[v] = cv.view(rows[v], K)
            if nxt_pro[0] is not None:
                raw, pros = c.view(tot, K), nxt_pro
            else:  # the layer's output itself feeds the next layer (no BatchNorm + ReLU in between)
                if outs[0].data_ptr() == c.data_ptr():
                    raw = cur = c.view(tot, K)
                else:
                    raw = cur = torch.empty(tot, K, dtype=dtype, device=dev)
                    for v in range(2):
                        kn.copy2d(outs[v], 0, K, cur, offs[v] * K, K, rows[v], K)
                pros = [None, None]
        if outs[0] is None:
            raise NotImplementedError("a head must end in Linear or BatchNorm (no trailing ReLU)")
        return [ChainRec(units[v], outs[v].view(rows[v], -1)) for v in range(2)]

    def chain_backward(self, rec: ChainRec, d_out: torch.Tensor, grads: GradStore, dtype, need_dx=True):
        """d_out: engine-owned buffer (overwritten in place)."""
        cur = d_out
        dev = d_out.device
        for i in range(len(rec.units) - 1, -1, -1):
            u = rec.units[i]
            if u.bn is not None:
                Cn = u.c.shape[-1]
                s2 = kn.new_stats(Cn, 2, dev)
                c2 = u.c.view(-1, Cn)
                cur2 = cur.view(-1, Cn)
                if u.relu:
                    kn.act_bwd_reduce(cur2, c2, u.st.scale, u.st.shift, cur2, s2)
                else:
                    kn.act_bwd_reduce(cur2, c2, None, None, None, s2)
                k = self._bn_bwd_coeffs(s2, 2, 1, u.bn, u.st, grads)
                kn.bn_bwd_apply(cur2, c2, k[0], k[1], k[2], cur2)
            self._unit_wgrad(u, cur, grads, dtype)
            if i > 0 or need_dx:
                cur = self._unit_dgrad(u, cur, dtype)
        return cur.view(cur.shape[0], -1) if need_dx else None

    def chain_backward_pair(self, recs, d_outs, grads: GradStore, dtype):
        """chain_backward for the two views of one head at once: BatchNorm backward and the input gradient per view
        (separate BatchNorm batches, backbone.py:140-145), but ONE weight-gradient launch per layer on the stacked
        rows -- the 18432-wide fuser layers have 256 rows per view and their weight gradient is a read-modify-write
        of a 1.4 GB fp32 matrix: once instead of twice."""
        curs = list(d_outs)
        dev = curs[0].device
        nun = len(recs[0].units)
        for i in range(nun - 1, -1, -1):
            us = [r.units[i] for r in recs]
            u0 = us[0]
            rows = [u.desc.N for u in us]
            Cin, Kout = u0.desc.C, u0.desc.K
            xcat = torch.empty(sum(rows), 1, 1, Cin, dtype=dtype, device=dev)
            dcat = torch.empty(sum(rows), 1, 1, Kout, dtype=dtype, device=dev)
            ks2 = None
            if u0.bn is not None:  # both views' backward sums first: their exchange is ONE message (_bn_bwd_coeffs_views)
                sums2 = []
                for v, u in enumerate(us):
                    Cn = u.c.shape[-1]
                    s2 = kn.new_stats(Cn, 2, dev)
                    c2 = u.c.view(-1, Cn)
                    cur2 = curs[v].view(-1, Cn)
                    if u.relu:
                        kn.act_bwd_reduce(cur2, c2, u.st.scale, u.st.shift, cur2, s2)
                    else:
                        kn.act_bwd_reduce(cur2, c2, None, None, None, s2)
                    sums2.append(s2)
                ks2 = self._bn_bwd_coeffs_views(sums2, 2, 1, u0.bn, [u.st for u in us], grads)
            off = 0
            for v, u in enumerate(us):
                dslot = dcat[off:off + rows[v]].view(rows[v], Kout)
                if u.bn is not None:
                    Cn = u.c.shape[-1]
                    k = ks2[v]
                    kn.bn_bwd_apply(curs[v].view(-1, Cn), u.c.view(-1, Cn), k[0], k[1], k[2], dslot)  # lands in its rows of the stacked gradient
                else:
                    kn.copy2d(curs[v], 0, Kout, dcat, off * Kout, Kout, rows[v], Kout)
                curs[v] = dslot
                xs = xcat[off:off + rows[v]]
                if u.x_pro is not None:
                    kn.bn_act(u.x, u.x_pro.scale, u.x_pro.shift, xs, relu=True)
                else:
                    kn.copy2d(u.x, 0, Cin, xcat, off * Cin, Cin, rows[v], Cin)
                off += rows[v]
            dpair = kn.conv_desc(dtype, sum(rows), 1, 1, Cin, Kout, 1, 1, 1, 0)
            if self.store_head_wgrad and sum(rows) <= 1024 and Kout * Cin >= (1 << 22):
                # few rows, a big matrix, and this is its ONLY launch of the step: the gradient is stored, not added --
                # no atomics, no read-modify-write of up to 1.36 GB of fp32, and the trainer's clear skips the tensor
                kn.conv_wgrad_store(dpair, xcat, dcat, grads.get(u0.op.weight))
                if hasattr(grads, "mark_stored"):
                    grads.mark_stored(u0.op.weight)
            else:
                if hasattr(grads, "unmark_stored"):
                    grads.unmark_stored(u0.op.weight)  # stored on an earlier step: holds stale values, not zeros
                kn.conv_wgrad(dpair, xcat, dcat, grads.get(u0.op.weight))
            bias = getattr(u0.op, "bias", None)
            if bias is not None:
                cs = torch.zeros(Kout, dtype=torch.float64, device=dev)
                kn.colsum(dcat, cs)
                kn.add_f64_to_f32(cs, grads.get(bias), 1.0)
            # ... and ONE input-gradient launch on the stacked rows (the weights are read once for both views)
            dxcat = self._unit_dgrad(Unit(u0.op, None, False, dpair, xcat, None, None), dcat, dtype)
            off = 0
            for v in range(len(us)):
                curs[v] = dxcat[off:off + rows[v]].view(rows[v], Cin)
                off += rows[v]
        return [c.view(c.shape[0], -1) for c in curs]

    # ---- whole model -----------------------------------------------------------------------------
    def model_forward(self, model: nn.Module, x1, x2, jigsaw_idx, dtype,
                      need_backward: bool = True) -> Tuple[tuple, StepRec]:
        B = x1[0].shape[0]
        K, n_keep = model.K, model.n_keep
        if x1[1].shape[0] != B * K or x2[1].shape[0] != B * K or x2[0].shape[0] != B:
            raise ValueError("target batches must hold B*K tiles")
        if jigsaw_idx is None or len(jigsaw_idx) != 2:
            raise ValueError("jigsaw_idx must be the two [B,K] index tensors")
        dev = x1[0].device
        if dev.type != "cuda":
            raise _lib.MsfwsiHipError("the MSF-WSI model runs only on a HIP device (no CPU path)")
        rec = StepRec(B)
        for v, idx in enumerate(jigsaw_idx):
            # the only in-path assertion of the reference (backbone.py:152)
            assert tuple(idx.shape) == (B, K), f"jigsaw_idx[{v}] must be [B,K]"
            rec.idx.append(idx.to(device=dev, dtype=torch.int64, non_blocking=True).contiguous())
        # reference call order (backbone.py:140-145): separate BatchNorm batches per call
        self._drop_c3 = False  # the first (small) context pass keeps everything and calibrates the planner
        shape_key = (tuple(x1[0].shape[1:]), tuple(x1[1].shape[1:]), sum(1 for _ in model.parameters()), str(dtype))
        multi = self._world() > 1 or (self.force_sync and dist.is_available() and dist.is_initialized())
        if self.dual_stream is not None:
            dual = self.dual_stream
        else:
            # automatic: backward wanted, and the previous step of this shape found the memory for it.  With several ranks
            # the finding is part of the COLLECTIVE plan (every rank takes the same branch) and the side streams exchange
            # their BatchNorm statistics on communicators of their own (MSFWSI_MULTIRANK_STREAMS=0: lockstep views instead)
            dual = (need_backward and self.allow_multistream and (not multi or self.multirank_streams)
                    and self._dual_ok.get((B, K) + shape_key, False))
        if dual and multi:
            self._make_stream_groups(dev)
        main = torch.cuda.current_stream(dev)
        side = self._side_stream(dev) if dual else None
        ev_c, ev_t = {}, {}
        if dual:
            self._prewarm((model.context_encoder, model.target_encoder), dtype, dev)
            side.wait_stream(main)  # inputs, weights, the zero arena: everything enqueued so far
        m0 = torch.cuda.memory_allocated(dev)
        self._bn_order = ("lead", ev_c) if dual else None
        # MSFWSI(..., use_checkpoint=True) (the reference's --use-ac, backbone.py:103-127): trade compute for activation
        # memory -- here: both target passes (16/17 of the images) run features-only and are re-run before their backward
        self._mode_override = "targets" if getattr(model, "use_checkpoint", False) else None
        lock = (not dual) and self._lockstep()
        c1_done = False
        ckey = (B, K) + shape_key
        tri = dual and need_backward and self.ctx_stream and ckey in self._calib
        third = self._side_stream(dev, "ctx") if tri else None
        if dual and self.dual_stream is None and ckey not in self._dual_started:
            # the automatic switch from one stream (first step of a shape) to several: the main stream's pool still caches
            # the blocks of ALL passes of that step, the side streams' pools would grow beside it (measured: 287 GiB
            # reserved and free-and-retry stalls instead of 257 GiB) -- hand the cached blocks back once
            self._dual_started.add(ckey)
            torch.cuda.empty_cache()
        if tri:
            # both context passes on the third stream, enqueued first; the plan comes from the calibration of an earlier
            # step (this step's first context pass is not measured alone), with both context passes still to come
            per_image, c3_frac = self._calib[ckey]
            nosave = self._plan_recompute(per_image, B, K, dev, c3_frac, shape_key, ctx_passes=2)
            third.wait_stream(main)
            for t in (x1[0], x2[0]):
                t.record_stream(third)
            self._bn_order = None  # one stream: view 0's running-statistics updates precede view 1's by stream order
            with torch.cuda.stream(third):
                rec.enc["c0"] = self.encoder_forward(model.context_encoder, x1[0], dtype)
                rec.enc["c1"] = self.encoder_forward(model.context_encoder, x2[0], dtype, save="c1" not in nosave)
            c1_done = True
        elif lock and need_backward and self._plan_known(B, K, shape_key):
            # the collective plan of this shape is known (second step on): the context views run as a lockstep pair too
            nosave = self._plan_recompute(0.0, B, K, dev, 0.0, shape_key)
            rec.enc["c0"], rec.enc["c1"] = self._run_views(
                lambda: self.encoder_forward(model.context_encoder, x1[0], dtype),
                lambda: self.encoder_forward(model.context_encoder, x2[0], dtype, save="c1" not in nosave))
            c1_done = True
        else:
            rec.enc["c0"] = self.encoder_forward(model.context_encoder, x1[0], dtype)
            per_image = (torch.cuda.memory_allocated(dev) - m0) / max(1, B)
            c3_bytes = sum(b.units[-1].c.numel() * b.units[-1].c.element_size() for b in rec.enc["c0"].blocks
                           if len(b.units) == 3 and b.units[-1].c is not None)
            c3_frac = (c3_bytes / max(1, B)) / per_image if per_image > 0 else 0.0
            if need_backward:
                self._calib[ckey] = (per_image, c3_frac)  # this pass ran first and alone: nothing else allocated meanwhile
            nosave = (self._plan_recompute(per_image, B, K, dev, c3_frac, shape_key) if need_backward
                      else {"c1", "t0", "t1"})
        if need_backward:
            # two streams hold two sets of transients at once, like the lockstep backward: same memory condition
            self._dual_ok[(B, K) + shape_key] = self._pair_bwd and not nosave
            if dual:
                self.last_plan += ",dual-stream" + ("+context-stream" if tri else "")
            if lock:
                self.last_plan += ",views-lockstep" + ("" if self._pair_bwd else "(forward only)")
        if not need_backward:
            rec.enc["c0"] = EncPass(model.context_encoder, B, 0, 0, None, None, None, None, [], rec.enc["c0"].feats,
                                    saved=False)
        if lock:
            # cross-replica statistics: the two views of an encoder in lockstep, one message per BatchNorm for both (on
            # the first step of a shape the first context pass above ran alone: it calibrates the memory plan)
            if not c1_done:
                rec.enc["c1"] = self.encoder_forward(model.context_encoder, x2[0], dtype, save="c1" not in nosave)
            rec.enc["t0"], rec.enc["t1"] = self._run_views(
                lambda: self.encoder_forward(model.target_encoder, x1[1], dtype, save="t0" not in nosave),
                lambda: self.encoder_forward(model.target_encoder, x2[1], dtype, save="t1" not in nosave))
        elif not dual:
            rec.enc["c1"] = self.encoder_forward(model.context_encoder, x2[0], dtype, save="c1" not in nosave)
            rec.enc["t0"] = self.encoder_forward(model.target_encoder, x1[1], dtype, save="t0" not in nosave)
            rec.enc["t1"] = self.encoder_forward(model.target_encoder, x2[1], dtype, save="t1" not in nosave)
        else:
            try:
                for t in (x2[0], x2[1]):
                    t.record_stream(side)
                if not c1_done:
                    with torch.cuda.stream(side):
                        self._bn_order = ("follow", ev_c)
                        rec.enc["c1"] = self.encoder_forward(model.context_encoder, x2[0], dtype, save="c1" not in nosave)
                self._bn_order = ("lead", ev_t)
                rec.enc["t0"] = self.encoder_forward(model.target_encoder, x1[1], dtype, save="t0" not in nosave)
                with torch.cuda.stream(side):
                    self._bn_order = ("follow", ev_t)
                    rec.enc["t1"] = self.encoder_forward(model.target_encoder, x2[1], dtype, save="t1" not in nosave)
            finally:
                self._bn_order = None
            main.wait_stream(side)
            if tri:
                main.wait_stream(third)
            for name in ("c0", "c1", "t1") if tri else ("c1", "t1"):  # allocated in another stream's pool, read by the heads on this one
                for f in rec.enc[name].feats:
                    f.record_stream(main)
        rec.dual = dual
        rec.tri = tri
        self.last_shape = (B, K, shape_key)
        if self.before_heads is not None:
            self.before_heads()  # a trainer's hook: e.g. wait for the previous step's Adam pass over the heads' weights
        rec.nosave = nosave
        rec.pair_bwd = self._pair_bwd
        outs = {}
        # heads: context group on the context stream, target group on view 1's stream, fuser group here -- all three read
        # finished encoder features; what they allocate lives in their stream's pool and is consumed there in the backward
        hstream = {"context": third, "target": side, "inter": None} if (tri and self.heads_on_streams) else {}
        rec.head_streams = bool(hstream)
        if hstream:
            for st in (side, third):
                st.wait_stream(main)  # the joins above + the trainer's hook
            for v in range(2):
                for f in rec.enc[f"t{v}"].feats:
                    f.record_stream(side)
            for t in rec.idx:
                t.record_stream(side)
        for grp in ("context", "target", "inter"):
            with (torch.cuda.stream(hstream[grp]) if hstream.get(grp) is not None else contextlib.nullcontext()):
                proj = getattr(model, f"{grp}_projector")
                pred = getattr(model, f"{grp}_predictor")
                for s in range(4):
                    fs = []
                    for v in range(2):
                        cf = rec.enc[f"c{v}"].feats[s]
                        tf = rec.enc[f"t{v}"].feats[s]
                        Cs = cf.shape[-1]
                        if grp == "context":
                            f = cf
                        elif grp == "target":
                            f = torch.empty_like(tf)
                            kn.rows_permute(tf, rec.idx[v], f, B, K, Cs)
                            rec.tgt_sorted[(s, v)] = f
                        else:
                            D = (n_keep + 1) * Cs
                            f = torch.empty(B, D, dtype=dtype, device=dev)
                            kn.copy2d(cf, 0, Cs, f, 0, D, B, Cs)
                            kn.copy2d(tf, 0, K * Cs, f, Cs, D, B, n_keep * Cs)
                        fs.append(f)
                    if self.pair_head_fwd:
                        zrecs = self.chain_forward_pair(proj[s], fs, dtype)
                        precs = self.chain_forward_pair(pred[s], [z.out for z in zrecs], dtype)
                    else:
                        zrecs = [self.chain_forward(proj[s], f, dtype) for f in fs]
                        precs = [self.chain_forward(pred[s], z.out, dtype) for z in zrecs]
                    for v in range(2):
                        rec.heads[(grp, s, v)] = (zrecs[v], precs[v])
                        outs[(grp, s, v)] = (precs[v].out, zrecs[v].out)
                        if hstream.get(grp) is not None:  # read by the loss on the main stream
                            precs[v].out.record_stream(main)
                            zrecs[v].out.record_stream(main)
        if hstream:
            main.wait_stream(side)
            main.wait_stream(third)
        result = []
        for grp in ("context", "target", "inter"):
            result.append((tuple(outs[(grp, s, 0)][0] for s in range(4)), tuple(outs[(grp, s, 1)][0] for s in range(4)),
                           tuple(outs[(grp, s, 0)][1] for s in range(4)), tuple(outs[(grp, s, 1)][1] for s in range(4))))
        return tuple(result), rec

    def model_backward(self, model: nn.Module, rec: StepRec, dps: Dict[Tuple[str, int, int], torch.Tensor],
                       grads: GradStore, dtype, on_group_done=None):
        """dps[(group, scale, view)] = dLoss/dp (engine-owned, storage dtype).  Fills `grads`."""
        B, K, n_keep = rec.B, model.K, model.n_keep
        dcf = [[None] * 4 for _ in range(2)]
        dtf = [[None] * 4 for _ in range(2)]
        dev0 = next(iter(dps.values())).device
        main0 = torch.cuda.current_stream(dev0)
        hstream = {}
        if rec.head_streams:
            # the groups' backward on the streams their forward ran on (their records live in those pools); the fuser group
            # here.  Its feature gradients are ADDED to the other two groups', so it waits for them before its first add.
            hstream = {"context": self._side_stream(dev0, "ctx"), "target": self._side_stream(dev0)}
            for g2, st in hstream.items():
                st.wait_stream(main0)  # the loss produced the dps
                for (g3, _, _), t in dps.items():
                    if g3 == g2:
                        t.record_stream(st)
        joined = not hstream
        for grp in ("context", "target", "inter"):
            with (torch.cuda.stream(hstream[grp]) if grp in hstream else contextlib.nullcontext()):
                # the fuser heads from the widest scale down: its 18432-wide layers are 2/3 of all gradient bytes, and their
                # bucket of the gradient exchange (on_group_done("inter", part=...)) leaves first
                for s in (range(3, -1, -1) if grp == "inter" else range(4)):
                    pair = None
                    if self.pair_head_wgrad:
                        hz, hp = zip(*(rec.heads.pop((grp, s, v)) for v in range(2)))
                        dzs = self.chain_backward_pair(hp, [dps[(grp, s, v)] for v in range(2)], grads, dtype)
                        pair = self.chain_backward_pair(hz, [d.contiguous() for d in dzs], grads, dtype)
                    for v in range(2):
                        if pair is not None:
                            df = pair[v]
                        else:
                            zrec, prec = rec.heads.pop((grp, s, v))
                            dp = dps[(grp, s, v)]
                            dz = self.chain_backward(prec, dp, grads, dtype)
                            df = self.chain_backward(zrec, dz.contiguous(), grads, dtype)
                        Cs = rec.enc[f"c{v}"].feats[s].shape[-1]
                        if grp == "context":
                            dcf[v][s] = df
                        elif grp == "target":
                            out = torch.empty_like(df)
                            kn.rows_permute(df, rec.idx[v], out, B, K, Cs, scatter=True)
                            dtf[v][s] = out
                        else:
                            if not joined:
                                for st in hstream.values():
                                    main0.wait_stream(st)
                                for row in dcf + dtf:
                                    for t in row:
                                        t.record_stream(main0)
                                joined = True
                            D = (n_keep + 1) * Cs
                            kn.copy2d(df, 0, D, dcf[v][s], 0, Cs, B, Cs, accumulate=True)
                            kn.copy2d(df, Cs, D, dtf[v][s], 0, K * Cs, B, n_keep * Cs, accumulate=True)
                    if grp == "inter" and on_group_done is not None and self.bucket_inter:
                        # this scale's projector / predictor weight gradients are complete: their buckets may travel
                        on_group_done("inter", part=f"inter_predictor.{s}.")
                        on_group_done("inter", part=f"inter_projector.{s}.")
                if grp == "inter" and on_group_done is not None:
                    on_group_done("inter")  # whatever no bucket covered (nothing, unless bucketing is off)
        # saved passes first (frees their activations), then the features-only ones are re-materialised
        order = sorted((("t0", dtf[0]), ("t1", dtf[1])), key=lambda nd: not rec.enc[nd[0]].saved)
        # view 1 on the side stream again (its activations live in that stream's pool); both views add into the same
        # gradient accumulators -- every such add is an atomic -- which must exist before either stream starts
        dual = rec.dual and getattr(grads, "preallocated", False)
        dev = dtf[0][0].device
        main = torch.cuda.current_stream(dev)
        side = self._side_stream(dev) if dual else None

        tri = dual and rec.tri
        third = self._side_stream(dev, "ctx") if tri else None

        def run(name, df):
            st = third if (tri and name.startswith("c")) else (side if dual and name.endswith("1") else None)
            if st is not None:
                for d in df:
                    if d is not None:
                        d.record_stream(st)
                with torch.cuda.stream(st):
                    self.encoder_backward(self._materialise(rec.enc.pop(name), dtype), df, grads, dtype)
            else:
                self.encoder_backward(self._materialise(rec.enc.pop(name), dtype), df, grads, dtype)

        if dual:
            side.wait_stream(main)  # the heads' backward produced the feature gradients
        if tri:  # the context passes' backward on the third stream, enqueued first, beside the target passes'
            third.wait_stream(main)
            for name, df in (("c1", dcf[1]), ("c0", dcf[0])):
                run(name, df)
        # lockstep backward of a pair of passes (one SyncBatchNorm message per BatchNorm for both views): only when both
        # passes kept their activations -- a features-only pass re-runs its forward first and would issue forward
        # exchanges where its partner issues backward ones -- and only when the memory plan left room for two sets of
        # backward transients (rec.pair_bwd: part of the COLLECTIVE plan, every rank takes the same branch)
        pair_ok = (not dual) and self._lockstep() and rec.pair_bwd

        def run_pair(n0, d0, n1, d1):
            p0, p1 = rec.enc.pop(n0), rec.enc.pop(n1)
            self._run_views(lambda: self.encoder_backward(p0, d0, grads, dtype),
                            lambda: self.encoder_backward(p1, d1, grads, dtype))

        if pair_ok and rec.enc["t0"].saved and rec.enc["t1"].saved:
            run_pair("t0", dtf[0], "t1", dtf[1])
        else:
            for name, df in order:
                run(name, df)
        if dual:
            main.wait_stream(side)
        if on_group_done is not None:
            on_group_done("target")
        if tri:
            main.wait_stream(third)
        elif pair_ok and rec.enc["c0"].saved and rec.enc["c1"].saved:
            run_pair("c0", dcf[0], "c1", dcf[1])
        else:
            for name, df in (("c1", dcf[1]), ("c0", dcf[0])):
                run(name, df)
        if dual:
            main.wait_stream(side)
        if on_group_done is not None:
            on_group_done("context")


# ------------------------------------------------------------------------------------------------
# autograd bridges (the drop-in boundary)
# ------------------------------------------------------------------------------------------------
_default_engine: Optional[Engine] = None


def default_engine() -> Engine:
    global _default_engine
    if _default_engine is None:
        _lib.load()  # fail loudly before anything else if the HIP library is absent
        _default_engine = Engine()
    return _default_engine


def _as_engine_grad(g: Optional[torch.Tensor], like: torch.Tensor) -> torch.Tensor:
    if g is None:
        return torch.zeros_like(like)
    return g.to(dtype=like.dtype).contiguous().clone()


class _ModelFn(torch.autograd.Function):
    """Whole-model node: inputs = images + every parameter; outputs = 24 p (differentiable) + 24 z."""

    @staticmethod
    def forward(ctx, model, engine, dtype, x1c, x1t, x2c, x2t, idx1, idx2, *params):
        outs, rec = engine.model_forward(model, (x1c, x1t), (x2c, x2t), [idx1, idx2], dtype)
        ctx.model, ctx.engine, ctx.dtype, ctx.rec = model, engine, dtype, rec
        ctx.params_ref = params
        flat_p, flat_z = [], []
        for grp in outs:
            flat_p += list(grp[0]) + list(grp[1])
            flat_z += list(grp[2]) + list(grp[3])
        ctx.mark_non_differentiable(*flat_z)
        return tuple(flat_p + flat_z)

    @staticmethod
    def backward(ctx, *gouts):
        model, engine, dtype, rec = ctx.model, ctx.engine, ctx.dtype, ctx.rec
        if rec is None:
            raise RuntimeError("the MSF-WSI HIP node supports a single backward pass")
        ctx.rec = None
        dps = {}
        i = 0
        for grp in ("context", "target", "inter"):
            for v in range(2):
                for s in range(4):
                    like = rec.heads[(grp, s, v)][1].out
                    dps[(grp, s, v)] = _as_engine_grad(gouts[i], like)
                    i += 1
        grads = GradStore()
        engine.model_backward(model, rec, dps, grads, dtype)
        pgrads = tuple(grads.logical(p) for p in ctx.params_ref)
        return (None,) * 9 + pgrads


def msfwsi_apply(model: nn.Module, x1, x2, jigsaw_idx):
    eng = getattr(model, "_engine", None) or default_engine()
    dtype = eng.compute_dtype()
    params = [p for p in model.parameters()]
    need_grad = torch.is_grad_enabled() and any(p.requires_grad for p in params)
    if not need_grad:
        outs, _ = eng.model_forward(model, x1, x2, jigsaw_idx, dtype, need_backward=False)
        return outs
    if jigsaw_idx is None or len(jigsaw_idx) != 2:
        raise ValueError("jigsaw_idx must be the two [B,K] index tensors")
    flat = _ModelFn.apply(model, eng, dtype, x1[0], x1[1], x2[0], x2[1], jigsaw_idx[0], jigsaw_idx[1], *params)
    p, z = flat[:24], flat[24:]
    res = []
    for gi in range(3):
        res.append((tuple(p[gi * 8:gi * 8 + 4]), tuple(p[gi * 8 + 4:gi * 8 + 8]), tuple(z[gi * 8:gi * 8 + 4]),
                    tuple(z[gi * 8 + 4:gi * 8 + 8])))
    return tuple(res)


class _EncoderFn(torch.autograd.Function):
    """Stand-alone encoder node (ResNet.forward outside MSFWSI)."""

    @staticmethod
    def forward(ctx, enc, engine, dtype, x, *params):
        ps = engine.encoder_forward(enc, x, dtype)
        ctx.enc, ctx.engine, ctx.dtype, ctx.ps, ctx.params = enc, engine, dtype, ps, params
        return tuple(ps.feats)

    @staticmethod
    def backward(ctx, *gouts):
        ps = ctx.ps
        if ps is None:
            raise RuntimeError("the MSF-WSI HIP node supports a single backward pass")
        ctx.ps = None
        df = [_as_engine_grad(g, f) for g, f in zip(gouts, ps.feats)]
        grads = GradStore()
        ctx.engine.encoder_backward(ps, df, grads, ctx.dtype)
        return (None,) * 4 + tuple(grads.logical(p) for p in ctx.params)


def encoder_apply(enc: nn.Module, x: torch.Tensor):
    eng = getattr(enc, "_engine", None) or default_engine()
    dtype = eng.compute_dtype()
    params = [p for n, p in enc.named_parameters() if not n.startswith("fc.")]
    if torch.is_grad_enabled() and any(p.requires_grad for p in params):
        return _EncoderFn.apply(enc, eng, dtype, x, *params)
    return tuple(eng.encoder_forward(enc, x, dtype).feats)
