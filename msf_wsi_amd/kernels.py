"""Typed Python front of the C ABI: torch tensors in, kernel launches on the current HIP stream out.

Every wrapper checks shapes/dtypes/devices on the host before launching (a faulting kernel can reset
the GPU), then forwards raw device pointers.  Tensors are NHWC ("[M][C]") as described in
include/msfwsi_hip.h.  Nothing here computes on the CPU or through torch operators.
"""
from __future__ import annotations

import ctypes as C
import threading
from typing import Optional

import torch

from . import _lib
from ._lib import ConvDesc, DT_BF16, DT_F16, DT_F32

LOWP = {torch.bfloat16: DT_BF16, torch.float16: DT_F16}

NSHARD = 32  # replicas of every fp64 statistics accumulator (spreads memory-side atomics)


def dt_of(t: torch.Tensor) -> int:
    if t.dtype == torch.float32:
        return DT_F32
    if t.dtype == torch.bfloat16:
        return DT_BF16
    if t.dtype == torch.float16:
        return DT_F16
    raise TypeError(f"unsupported storage dtype {t.dtype}")


def vec_of(dtype: torch.dtype) -> int:
    return 4 if dtype == torch.float32 else 8


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _p(t: Optional[torch.Tensor]) -> Optional[int]:
    if t is None:
        return None
    return t.data_ptr()


def _req(t: torch.Tensor, name: str, dtype=None, numel=None):
    if not t.is_cuda:
        raise _lib.MsfwsiHipError(f"{name}: expected a GPU tensor (no CPU path exists)")
    if not t.is_contiguous():
        raise ValueError(f"{name}: must be contiguous")
    if dtype is not None and t.dtype != dtype:
        raise TypeError(f"{name}: expected {dtype}, got {t.dtype}")
    if numel is not None and t.numel() != numel:
        raise ValueError(f"{name}: expected {numel} elements, got {t.numel()}")
    if t.numel() > 1 and t.data_ptr() % 16 != 0:
        raise ValueError(f"{name}: must be 16-byte aligned")


def _opt(t, name, dtype=None, numel=None):
    if t is not None:
        _req(t, name, dtype, numel)


def conv_desc(dtype: torch.dtype, N, H, W, Cin, K, R, S, stride, pad) -> ConvDesc:
    P = (H + 2 * pad - R) // stride + 1
    Q = (W + 2 * pad - S) // stride + 1
    code = DT_F32 if dtype == torch.float32 else LOWP[dtype]
    return ConvDesc(code, N, H, W, Cin, P, Q, K, R, S, stride, pad)


class ZeroArena:
    """Zero-initialised scratch for one step.  The step needs ~6000 small zeroed accumulators (sharded BatchNorm
    statistics, fold matrices); instead of one fill kernel each, `begin_step` clears ONE buffer (sized from the
    previous step's demand) and `zeros` hands out 256-byte aligned slices of it.  Inactive (plain torch.zeros)
    outside PretrainStep.step and while the first step measures the demand."""

    def __init__(self):
        self.buf: Optional[torch.Tensor] = None
        self.off = 0
        self.demand = 0
        self.active = False
        self._lock = threading.Lock()  # the two view passes of an encoder may run on two host threads (engine._ViewPair)

    def begin_step(self, device):
        want = int(self.demand * 1.1) + (1 << 20)
        if self.demand and (self.buf is None or self.buf.numel() < want or self.buf.device != torch.device(device)):
            self.buf = torch.empty(want, dtype=torch.uint8, device=device)
        if self.buf is not None:
            self.buf.zero_()
        self.off, self.demand, self.active = 0, 0, True

    def end_step(self):
        self.active = False

    def zeros(self, shape, dtype, device) -> torch.Tensor:
        n = 1
        for d in shape:
            n *= int(d)
        nbytes = (n * torch.empty((), dtype=dtype).element_size() + 255) // 256 * 256
        if not self.active:
            return torch.zeros(shape, dtype=dtype, device=device)
        with self._lock:
            self.demand += nbytes
            if self.buf is None or self.off + nbytes > self.buf.numel() or self.buf.device != torch.device(device):
                off = -1
            else:
                off = self.off
                self.off += nbytes
        if off < 0:
            return torch.zeros(shape, dtype=dtype, device=device)
        return self.buf[off:off + nbytes].view(dtype)[:n].view(shape)


ARENA = ZeroArena()


def zeros(shape, dtype, device) -> torch.Tensor:
    return ARENA.zeros(tuple(shape), dtype, device)


def new_stats(C_: int, slots: int = 2, device=None) -> torch.Tensor:
    return ARENA.zeros((NSHARD, slots, C_), torch.float64, device or "cuda")


class KernelTimer:
    """Optional per-launch timing of the dense kernels with HIP events recorded on the launch stream
    (bench.py's roofline leg).  For every launch it keeps the algorithmic FLOPs and the algorithmic HBM
    bytes (each operand / result tensor counted once) next to the event pair."""

    def __init__(self, streams: bool = False):
        self.records = []  # (kind, symbol, flops, bytes, ev0, ev1, shape)
        self.stream_records = [] if streams else None  # (name, bytes, ev0, ev1): the HBM-bound elementwise kernels

    def by_shape(self):
        """per (kernel family, layer shape): launches, seconds, algorithmic FLOPs and bytes -- the layer report"""
        torch.cuda.synchronize()
        agg = {}
        for kind, sym, fl, by, e0, e1, shape in self.records:
            a = agg.setdefault((kind, shape, sym), [0, 0.0, 0.0, 0.0])
            a[0] += 1
            a[1] += e0.elapsed_time(e1) * 1e-3
            a[2] += fl
            a[3] += by
        for name, by, e0, e1 in self.stream_records or []:
            a = agg.setdefault(("stream", name, name), [0, 0.0, 0.0, 0.0])
            a[0] += 1
            a[1] += e0.elapsed_time(e1) * 1e-3
            a[3] += by
        return agg

    def summary(self, by_symbol: bool = False):
        """per kernel family (conv_fwd / conv_dgrad / conv_wgrad) or, by_symbol, per kernel symbol (the template
        instance the C side dispatches to -- the name rocprofv3 --kernel-trace --stats reports)"""
        torch.cuda.synchronize()
        agg = {}
        for kind, sym, fl, by, e0, e1, _shape in self.records:
            a = agg.setdefault(sym if by_symbol else kind, [0, 0.0, 0.0, 0.0, kind])
            a[0] += 1
            a[1] += e0.elapsed_time(e1) * 1e-3
            a[2] += fl
            a[3] += by
        if not by_symbol and self.stream_records:  # the HBM-bound elementwise kernels as one family (streams=True)
            a = agg.setdefault("stream", [0, 0.0, 0.0, 0.0, "stream"])
            for _name, by, e0, e1 in self.stream_records:
                a[0] += 1
                a[1] += e0.elapsed_time(e1) * 1e-3
                a[3] += by
        return {k: {"launches": v[0], "seconds": v[1], "flops": v[2], "bytes": v[3], "family": v[4]}
                for k, v in agg.items()}


TIMER: Optional[KernelTimer] = None
_TIMER_LOCK = threading.Lock()


_TCODE = {4: "f", 2: "DF16b"}  # Itanium codes of float / __bf16 (fp16 "DF16_" is set by the caller's dtype)


def _tuning(key: int) -> int:
    """the library's current value of a dispatch switch (msfwsi_get_tuning): _symbol mirrors the C side's dispatch from the
    SAME values, whoever set them (MSFWSI_TUNING at load time, a test's helpers.tuned)"""
    v = C.c_long()
    _lib.check(_lib.load().msfwsi_get_tuning(int(key), C.byref(v)), "msfwsi_get_tuning")
    return int(v.value)



def _symbol(kind: str, d: ConvDesc, tcode: str, pro: bool, halo: bool = False, epi: int = 0, two: bool = False,
            p2: bool = False) -> str:
    """The mangled template-instance fragment of the kernel this launch reaches; mirrors dispatch_tile() /
    launch_igemm() (csrc/igemm.hip), msfwsi_conv_wgrad() (csrc/wgrad.hip) and conv3x3.hip.  Only used to label
    timings so that bench.py's roofline names the same kernel as the rocprofv3 summary."""
    es = 4 if tcode == "f" else 2
    bk = 16 if es == 4 else 32
    if kind == "conv_wgrad":
        if (es == 2 and d.C == 16 and d.K == 64 and (d.R, d.S, d.stride, d.pad) == (4, 4, 1, 2) and d.P == d.H
                and d.N * (d.H + 2) * (d.W + 2) >= 32 * 512 * 256 and _tuning(12) != 0):
            return f"stem_wgrad_os_kernelI{tcode}Lb0EE"
        if (es == 2 and d.C == 64 and d.K == 64 and d.R == 3 and d.S == 3 and d.stride == 1 and d.pad == 1
                and d.N * (d.H + 1) * (d.W + 1) >= 32 * 256 * 256 and _tuning(10) != 0):
            return f"wgrad_os_kernelI{tcode}E"
        bi = 64 if d.K <= 64 else 128
        bj = 64 if d.R * d.S * d.C <= 64 else 128
        if (es == 2 and not pro and d.K % 256 == 0 and (d.R * d.S * d.C) % 256 == 0
                and _tuning(6) != 0):
            bi = bj = 256  # the 16-wave tile of the deep layers (msfwsi_conv_wgrad)
        lin = (not pro and _tuning(2) != 0 and d.stride == 1 and d.P == d.H
               and d.Q == d.W and d.pad <= 1 and d.R <= 3 and d.S <= 3 and d.R == 2 * d.pad + 1 and d.S == 2 * d.pad + 1)
        return f"wgrad_kernelI{tcode}Li{bi}ELi{bj}ELb{int(pro)}ELb{int(lin)}E"
    dgrad = kind == "conv_dgrad"
    if halo:
        if es == 2 and d.C == 64 and d.K == 64 and _tuning(9) != 0:
            return f"conv3x3_ws_kernelI{tcode}Lb{int(dgrad)}E"
        bn = 64 if (d.C if dgrad else d.K) <= 64 else 128
        return f"conv3x3_kernelI{tcode}Li{bn}ELb{int(dgrad)}E"
    M = d.N * (d.H * d.W if dgrad else d.P * d.Q)
    nout, csrc = (d.C, d.K) if dgrad else (d.K, d.C)
    if nout <= 64:
        tile = (128, 64, 2, 2)
    elif not pro and es == 2 and ((M + 255) // 256) * ((nout + 127) // 128) >= _tuning(0):
        tile = (256, 128, 4, 2)
    elif not pro and ((M + 127) // 128) * ((nout + 127) // 128) < _tuning(4):
        tile = (128, 64, 2, 2)
    else:
        tile = (128, 128, 2, 2)
    t = "Li%dELi%dELi%dELi%dE" % tile
    if not pro and _tuning(1) != 0 and csrc % bk == 0 and d.R * d.S <= 32:
        return f"igemm_dma_kernelI{tcode}{t}Lb{int(dgrad)}ELi{epi}ELb{int(two)}ELb0ELb{int(p2)}E"
    return f"igemm_kernelI{tcode}{t}Lb{int(dgrad)}ELb{int(pro)}E"


def _timed(kind, d: ConvDesc, esize: int, fn, extra_elems: int = 0, pro: bool = False, halo: bool = False,
           dtype=None, symbol_override: Optional[str] = None, epi: int = 0, two: bool = False, extra_k: int = 0,
           same_operand: bool = False, p2: bool = False):
    """extra_elems: elements of the additional activation-sized operands the launch reads (residual / identity, the
    gate's activation, the second source of a two-source launch) -- algorithmic bytes of the fused work, counted once
    each.  extra_k: the second source's reduction range (its FLOPs and weight bytes).  same_operand: a Gram launch
    conv_wgrad(d, a, a, A) reads ONE tensor (counted once; it was counted twice until round 3, which put the 56x56
    64x64 Gram row of the layer report above the HBM peak)"""
    if TIMER is None:
        return fn()
    M = d.N * d.P * d.Q
    nout, rows = (d.C, d.N * d.H * d.W) if kind == "conv_dgrad" else (d.K, M)
    flops = 2.0 * M * d.K * d.R * d.S * d.C + 2.0 * rows * nout * extra_k
    act_in = 0 if same_operand else d.N * d.H * d.W * d.C
    nbytes = float(esize) * (act_in + M * d.K + extra_elems) + float(
        esize if kind != "conv_wgrad" else 4) * (d.K * d.R * d.S * d.C + nout * extra_k)
    tcode = "DF16_" if dtype == torch.float16 else _TCODE[esize]
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with _TIMER_LOCK:  # two view passes in lockstep share the stream: the pair must bracket exactly this launch
        e0.record()
        r = fn()
        e1.record()
    shape = f"N{d.N} {d.H}x{d.W} C{d.C}->K{d.K} {d.R}x{d.S}/s{d.stride}" + (" +src2" if two else "") + (
        f" epi{epi}" if epi else "") + (f" +{extra_elems * esize >> 20}MiB epilogue" if extra_elems else "")
    TIMER.records.append((kind, symbol_override or _symbol(kind, d, tcode, pro, halo, epi, two, p2), flops, nbytes, e0, e1,
                          shape))
    return r


def _stream_timed(name: str, nbytes: float, fn):
    """the HBM-bound elementwise kernels (KernelTimer(streams=True): the layer report and the `stream` family of the bench line)"""
    if TIMER is None or TIMER.stream_records is None:
        return fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with _TIMER_LOCK:
        e0.record()
        r = fn()
        e1.record()
    TIMER.stream_records.append((name, float(nbytes), e0, e1))
    return r


# ------------------------------------------------------------------------------------------------
def conv_fwd(d: ConvDesc, x, w, y, pro=None, bias=None, stats=None):
    """y = conv(act(x), w) (+bias); pro = (scale, shift) of the producer BatchNorm; stats [NSHARD,2,K]."""
    lib = _lib.load()
    dt = x.dtype
    _req(x, "x", dt, d.N * d.H * d.W * d.C)
    _req(w, "w", dt, d.K * d.R * d.S * d.C)
    _req(y, "y", dt, d.N * d.P * d.Q * d.K)
    ps = psh = None
    if pro is not None:
        ps, psh = pro
        _req(ps, "pro_scale", torch.float32, d.C)
        _req(psh, "pro_shift", torch.float32, d.C)
    _opt(bias, "bias", torch.float32, d.K)
    nsh = 1
    if stats is not None:
        _req(stats, "stats", torch.float64)
        nsh = stats.shape[0]
        if stats.numel() != nsh * 2 * d.K:
            raise ValueError("stats must be [nshard,2,K]")
    _timed("conv_fwd", d, x.element_size(), lambda: _lib.check(
        lib.msfwsi_conv_fwd(C.byref(d), _p(x), _p(w), _p(y), _p(ps), _p(psh), _p(bias), _p(stats), nsh, _stream()),
        "conv_fwd"), pro=pro is not None, dtype=dt)
    return y


def gate_numel(rows: int, Cn: int, dtype) -> int:
    """bytes of the gate-byte tensor of a [rows][Cn] activation (csrc/common.h gate_off: rows padded to 128 in the blocked
    form, which every width with whole dwords of gate bytes per row uses)"""
    cpr = Cn // vec_of(dtype)
    return int(rows) * cpr if cpr % 4 else (int(rows) + 127) // 128 * 128 * cpr


def gate_bytes(d_or_rows, Cn: int = 0, dtype=None, device="cuda") -> torch.Tensor:
    """flat uint8 buffer for the ReLU-gate bits of a [rows][C] activation: one byte per 16-byte chunk (vec = 4 fp32 /
    8 16-bit elements), laid out by gate_off (csrc/common.h); gate_unpack / gate_pack convert from / to [rows][C/vec]"""
    return torch.empty(gate_numel(int(d_or_rows), Cn, dtype), dtype=torch.uint8, device=device)


def _gate_index(rows: int, cpr: int, device) -> torch.Tensor:
    m = torch.arange(rows, device=device, dtype=torch.int64).view(-1, 1)
    c = torch.arange(cpr, device=device, dtype=torch.int64).view(1, -1)
    if cpr % 4:
        return m * cpr + c
    return ((m >> 7) * (cpr >> 2) + (c >> 2)) * 512 + (m & 127) * 4 + (c & 3)


def gate_unpack(buf: torch.Tensor, rows: int, Cn: int, dtype) -> torch.Tensor:
    """the gate bytes as a [rows][C/vec] uint8 tensor (tests / diagnostics: plain torch indexing)"""
    return buf.view(-1)[_gate_index(rows, Cn // vec_of(dtype), buf.device)]


def gate_pack(linear: torch.Tensor, Cn: int, dtype) -> torch.Tensor:
    """[rows][C/vec] uint8 -> a gate-byte buffer in the kernels' layout (tests)"""
    rows = linear.shape[0]
    buf = torch.zeros(gate_numel(rows, Cn, dtype), dtype=torch.uint8, device=linear.device)
    buf[_gate_index(rows, Cn // vec_of(dtype), linear.device)] = linear.to(torch.uint8)
    return buf


def stem_conv_fwd(x, w_run, y, stats, R, S, stride, pad, P: int = 0, Q: int = 0) -> bool:
    """the stem conv as R row taps over runs of S contiguous pixels (x [N,H,W,CP], w_run [K][R][run], run = S*CP
    padded to whole k slabs with zero columns); P, Q: explicitly cropped output extent (0 = formula); False if the
    library has no kernel for the shape"""
    lib = _lib.load()
    N, H, W, CP = x.shape
    K = y.shape[-1]
    _req(x, "x")
    _req(w_run, "w_run", x.dtype)
    _req(y, "y", x.dtype)
    nsh = 1
    if stats is not None:
        _req(stats, "stats", torch.float64)
        nsh = stats.shape[0]
    d = conv_desc(x.dtype, N, H, W, CP, K, R, S, stride, pad)
    if P or Q:
        d = ConvDesc(d.dtype, N, H, W, CP, int(P), int(Q), K, R, S, stride, pad)
    if y.numel() != N * d.P * d.Q * K:
        raise ValueError(f"stem_conv_fwd: y must hold {N}x{d.P}x{d.Q}x{K} elements")
    rc = [0]

    def run():
        rc[0] = lib.msfwsi_stem_conv_fwd(dt_of(x), _p(x), _p(w_run), _p(y), _p(stats), nsh, N, H, W, CP, K, R, S,
                                         stride, pad, int(P), int(Q), _stream())
        if rc[0] != -2:
            _lib.check(rc[0], "stem_conv_fwd")

    tc = "DF16_" if x.dtype == torch.float16 else _TCODE[x.element_size()]
    ws = (x.element_size() == 2 and CP == 16 and K == 64 and (R, S, stride, pad) == (4, 4, 1, 2) and d.P == H and d.Q == W
          and W <= 128 and _tuning(12) != 0)  # mirrors msfwsi_stem_ws_fwd (csrc/stem.hip)
    _timed("conv_fwd", d, x.element_size(), run, dtype=x.dtype,
           symbol_override=f"stem_ws_kernelI{tc}E" if ws else f"igemm_dma_kernelI{tc}Li128ELi64ELi2ELi2ELb0ELi0ELb0ELb1ELb0E")
    return rc[0] == 0


def conv_fwd_post(d: ConvDesc, x, w, y, post_scale, post_shift, ident=None, relu=True, gate_out=None):
    """y = [relu](round(conv(x, w)) * post_scale + post_shift + ident): conv with its consumer BatchNorm (statistics
    known beforehand), the residual add and the ReLU in the epilogue"""
    lib = _lib.load()
    dt = x.dtype
    _req(x, "x", dt, d.N * d.H * d.W * d.C)
    _req(w, "w", dt, d.K * d.R * d.S * d.C)
    _req(y, "y", dt, d.N * d.P * d.Q * d.K)
    _req(post_scale, "post_scale", torch.float32, d.K)
    _req(post_shift, "post_shift", torch.float32, d.K)
    _opt(ident, "ident", dt, d.N * d.P * d.Q * d.K)
    _opt(gate_out, "gate_out", torch.uint8, gate_numel(d.N * d.P * d.Q, d.K, dt))
    _timed("conv_fwd", d, x.element_size(), lambda: _lib.check(
        lib.msfwsi_conv_fwd_post(C.byref(d), _p(x), _p(w), _p(y), _p(post_scale), _p(post_shift), _p(ident),
                                 int(bool(relu)), _p(gate_out), _stream()), "conv_fwd_post"),
        extra_elems=ident.numel() if ident is not None else 0, dtype=dt, epi=1)
    return y


def conv_fwd_post2(d: ConvDesc, x, w_cat, y, src2, post_scale, post_shift, ident=None, relu=True, gate_out=None) -> bool:
    """y = [relu](round(x . w_cat[:, :C] + src2 . w_cat[:, C:]) * post_scale + post_shift + ident); False if the
    library has no kernel for the shape"""
    lib = _lib.load()
    dt = x.dtype
    C2 = src2.shape[-1]
    _req(x, "x", dt, d.N * d.H * d.W * d.C)
    _req(w_cat, "w_cat", dt, d.K * (d.C + C2))
    _req(y, "y", dt, d.N * d.P * d.Q * d.K)
    _req(src2, "src2", dt, d.N * d.P * d.Q * C2)
    _req(post_scale, "post_scale", torch.float32, d.K)
    _req(post_shift, "post_shift", torch.float32, d.K)
    _opt(ident, "ident", dt, d.N * d.P * d.Q * d.K)
    _opt(gate_out, "gate_out", torch.uint8, gate_numel(d.N * d.P * d.Q, d.K, dt))
    rc = [0]

    def run():
        rc[0] = lib.msfwsi_conv_fwd_post2(C.byref(d), _p(x), _p(w_cat), _p(y), _p(src2), int(C2), _p(post_scale),
                                          _p(post_shift), _p(ident), int(bool(relu)), _p(gate_out), _stream())
        if rc[0] != -2:
            _lib.check(rc[0], "conv_fwd_post2")

    _timed("conv_fwd", d, x.element_size(), run, extra_elems=src2.numel() + (ident.numel() if ident is not None else 0),
           dtype=dt, epi=1, two=True, extra_k=int(C2))
    return rc[0] == 0


def row_scale_cat(W1, s1, W2, s2, b1, b2, out, shift):
    """out = [s1 o W1 | s2 o W2] (rows scaled, columns concatenated), shift = b1 + b2   (all fp32)"""
    lib = _lib.load()
    K = s1.numel()
    C1, C2 = W1.numel() // K, W2.numel() // K
    _req(W1, "W1", torch.float32, K * C1)
    _req(W2, "W2", torch.float32, K * C2)
    for nm, t in (("s1", s1), ("s2", s2), ("b1", b1), ("b2", b2), ("shift", shift)):
        _req(t, nm, torch.float32, K)
    _req(out, "out", torch.float32, K * (C1 + C2))
    _lib.check(lib.msfwsi_row_scale_cat(_p(W1), _p(s1), C1, _p(W2), _p(s2), C2, _p(b1), _p(b2), _p(out), _p(shift), K,
                                        _stream()), "row_scale_cat")


def pixel_stride(x, out, stride: int, expand: bool):
    """expand=False: out[N,P,Q,C] = x[:, ::stride, ::stride, :]; expand=True: out[N,H,W,C] = x[N,P,Q,C] zero-stuffed"""
    lib = _lib.load()
    full, lo = (out, x) if expand else (x, out)
    N, H, W, Cn = full.shape
    P, Q = (H - 1) // stride + 1, (W - 1) // stride + 1
    _req(x, "x")
    _req(out, "out", x.dtype)
    if tuple(lo.shape) != (N, P, Q, Cn):
        raise ValueError(f"pixel_stride: low-resolution tensor must be {(N, P, Q, Cn)}, got {tuple(lo.shape)}")
    _stream_timed("pixel_stride", x.element_size() * (x.numel() if expand else out.numel()) * 2, lambda: _lib.check(
        lib.msfwsi_pixel_stride(dt_of(x), _p(x), _p(out), N, H, W, Cn, int(stride), int(bool(expand)), _stream()),
        "pixel_stride"))
    return out


def fold_matvec(W, v, out):
    """out[k] = sum_c W[k][c] * v[c]   (fp64)"""
    lib = _lib.load()
    K = out.numel()
    Cn = v.numel()
    _req(W, "W", torch.float32, K * Cn)
    _req(v, "v", torch.float64, Cn)
    _req(out, "out", torch.float64, K)
    _lib.check(lib.msfwsi_fold_matvec(_p(W), _p(v), _p(out), K, Cn, _stream()), "fold_matvec")


def conv_dgrad(d: ConvDesc, dy, w, dx, resid=None, gapg=None, gap_scale=0.0, mask=None, sums=None, mask_bits=None,
               resid_stride=1):
    """mask = (c, scale, shift) of the activation that produced the conv input: fuses its ReLU gate and the
    BatchNorm-backward sums {sum g, sum g*c} (-> sums [nshard,2,C]) into the epilogue.  mask_bits: the gate as
    the bytes conv_fwd_post wrote (sums slot 0 only)."""
    lib = _lib.load()
    dt = dy.dtype
    _req(dy, "dy", dt, d.N * d.P * d.Q * d.K)
    _req(w, "w", dt, d.K * d.R * d.S * d.C)
    _req(dx, "dx", dt, d.N * d.H * d.W * d.C)
    if resid_stride > 1:  # low-resolution residual, added on the strided sub-grid only
        _req(resid, "resid", dt, d.N * ((d.H - 1) // resid_stride + 1) * ((d.W - 1) // resid_stride + 1) * d.C)
    else:
        _opt(resid, "resid", dt, d.N * d.H * d.W * d.C)
    _opt(gapg, "gapg", dt, d.N * d.C)
    mc = msc = msh = None
    nsh = 1
    if mask is not None:
        mc, msc, msh = mask
        _req(mc, "mask_c", dt, d.N * d.H * d.W * d.C)
        _req(msc, "mask_scale", torch.float32, d.C)
        _req(msh, "mask_shift", torch.float32, d.C)
        _req(sums, "sums", torch.float64)
        nsh = sums.numel() // (2 * d.C)
        if nsh * 2 * d.C != sums.numel():
            raise ValueError("sums must be [nshard,2,C]")
    elif mask_bits is not None:
        _req(mask_bits, "mask_bits", torch.uint8, gate_numel(d.N * d.H * d.W, d.C, dt))
        _req(sums, "sums", torch.float64)
        nsh = sums.numel() // (2 * d.C)
        if nsh * 2 * d.C != sums.numel():
            raise ValueError("sums must be [nshard,2,C]")
    elif sums is not None:
        raise ValueError("sums without mask")
    _timed("conv_dgrad", d, dy.element_size(), lambda: _lib.check(
        lib.msfwsi_conv_dgrad(C.byref(d), _p(dy), _p(w), _p(dx), _p(resid), _p(gapg), float(gap_scale), _p(mc),
                              _p(msc), _p(msh), _p(mask_bits), _p(sums), nsh, int(resid_stride), _stream()),
        "conv_dgrad"),
        extra_elems=(resid.numel() if resid is not None else 0) + (dx.numel() if mask is not None else 0)
        + (dx.numel() // 16 if mask_bits is not None else 0), dtype=dt, epi=3 if resid_stride > 1 else 0)
    return dx


# ---- activation-stationary ("panel") 1x1 kernels (csrc/panel.hip) -----------------------------------------------
def panel_supported(d: ConvDesc, dgrad: bool) -> bool:
    return bool(_lib.load().msfwsi_panel_supported(C.byref(d), int(bool(dgrad))))


def panel_pack_weights(w, wpk, nout: int, k: int, stride_n: int, stride_k: int):
    """wpk[n/32][k/16][64][8] <- W(n, k) = w.flat[n*stride_n + k*stride_k] (MFMA fragment order)"""
    lib = _lib.load()
    dt = w.dtype
    _req(w, "w", dt, nout * k)
    _req(wpk, "wpk", dt, nout * k)
    _lib.check(lib.msfwsi_panel_pack_weights(dt_of(w), _p(w), _p(wpk), int(nout), int(k), int(stride_n), int(stride_k),
                                             _stream()), "panel_pack_weights")
    return wpk


def panel_gram(c, scale, shift, A, sums) -> bool:
    """A[C][1][1][C] (fp32, zeroed) += a^T a and sums [C] (fp64, zeroed) += column sums of a = relu(scale*c + shift), in one
    pass over the raw conv output c [.., C] (bn_act_sum + gram without the normalised activation); False if the library
    has no kernel for the width"""
    lib = _lib.load()
    Cn = c.shape[-1]
    M = c.numel() // Cn
    _req(c, "c", c.dtype)
    _req(scale, "scale", torch.float32, Cn)
    _req(shift, "shift", torch.float32, Cn)
    _req(A, "A", torch.float32, Cn * Cn)
    _req(sums, "sums", torch.float64, Cn)
    if c.dtype == torch.float32:
        return False
    A64 = ARENA.zeros((Cn * Cn,), torch.float64, c.device)
    d = conv_desc(c.dtype, M, 1, 1, Cn, Cn, 1, 1, 1, 0)
    rc = _timed("conv_wgrad", d, c.element_size(), lambda: lib.msfwsi_panel_gram(
        dt_of(c), _p(c), _p(scale), _p(shift), _p(A64), _p(sums), M, Cn, _stream()), dtype=c.dtype, same_operand=True,
        symbol_override=f"panel_gram_kernelI{'DF16_' if c.dtype == torch.float16 else 'DF16b'}Li{Cn}ELi4EE")
    if rc == -2:
        return False
    _lib.check(rc, "panel_gram")
    add_f64_to_f32(A64, A)
    return True


def _panel_symbol(d: ConvDesc, dt, k: int, pro: int, epi: int, hand: bool) -> str:
    """the template-instance fragment rocprofv3 reports (mirrors launch_panel, csrc/panel.hip)"""
    tcode = "DF16_" if dt == torch.float16 else "DF16b"
    bm = 64 if k == 512 else 128
    hand = hand and (d.N * d.H * d.W) % bm == 0
    return f"panel_kernelI{tcode}Li{k}ELi{bm}ELi{pro}ELi{epi}ELb{int(hand)}E"


def panel_fwd_post(d: ConvDesc, x, wpk, y, post_scale, post_shift, pro=None, ident=None, relu=True, gate_out=None) -> bool:
    """y = [relu](round(act(x) . W^T) * post_scale + post_shift + ident), act = relu(pro_scale*x + pro_shift) when pro is
    given (x = the producer's raw conv output); False if the library has no panel kernel for the shape"""
    lib = _lib.load()
    dt = x.dtype
    _req(x, "x", dt, d.N * d.H * d.W * d.C)
    _req(wpk, "wpk", dt, d.K * d.C)
    _req(y, "y", dt, d.N * d.P * d.Q * d.K)
    _req(post_scale, "post_scale", torch.float32, d.K)
    _req(post_shift, "post_shift", torch.float32, d.K)
    ps = psh = None
    if pro is not None:
        ps, psh = pro
        _req(ps, "pro_scale", torch.float32, d.C)
        _req(psh, "pro_shift", torch.float32, d.C)
    _opt(ident, "ident", dt, d.N * d.P * d.Q * d.K)
    _opt(gate_out, "gate_out", torch.uint8, gate_numel(d.N * d.P * d.Q, d.K, dt))
    rc = _timed("conv_fwd", d, x.element_size(), lambda: lib.msfwsi_panel_fwd_post(
        C.byref(d), _p(x), _p(ps), _p(psh), _p(wpk), _p(y), _p(post_scale), _p(post_shift), _p(ident), int(bool(relu)),
        _p(gate_out), _stream()), extra_elems=ident.numel() if ident is not None else 0, dtype=dt, epi=1,
        symbol_override=_panel_symbol(d, dt, d.C, 1 if pro is not None else 0, 1, ident is not None and gate_out is not None))
    if rc == -2:
        return False
    _lib.check(rc, "panel_fwd_post")
    return True


def panel_dgrad(d: ConvDesc, dy, wpk, dx, bnbwd=None, dc_out=None, resid=None, resid_stride=1, gapg=None, gap_scale=0.0,
                mask_bits=None, sums=None) -> bool:
    """dx = gate(round(dc . W) + resid + gap_scale*gapg), sums slot 0 += dx; dc = k1*dy + k2*c + k3 when bnbwd = (c, k1,
    k2, k3) (written to dc_out if given), else dy.  False if the library has no panel kernel for the shape"""
    lib = _lib.load()
    dt = dy.dtype
    _req(dy, "dy", dt, d.N * d.P * d.Q * d.K)
    _req(wpk, "wpk", dt, d.K * d.C)
    _req(dx, "dx", dt, d.N * d.H * d.W * d.C)
    c = k1 = k2 = k3 = None
    if bnbwd is not None:
        c, k1, k2, k3 = bnbwd
        _req(c, "c", dt, d.N * d.P * d.Q * d.K)
        for nm, t in (("k1", k1), ("k2", k2), ("k3", k3)):
            _req(t, nm, torch.float32, d.K)
        _opt(dc_out, "dc_out", dt, d.N * d.P * d.Q * d.K)
    elif dc_out is not None:
        raise ValueError("dc_out without bnbwd")
    if resid_stride > 1:
        _req(resid, "resid", dt, d.N * ((d.H - 1) // resid_stride + 1) * ((d.W - 1) // resid_stride + 1) * d.C)
    else:
        _opt(resid, "resid", dt, d.N * d.H * d.W * d.C)
    _opt(gapg, "gapg", dt, d.N * d.C)
    nsh = 1
    if mask_bits is not None:
        _req(mask_bits, "mask_bits", torch.uint8, gate_numel(d.N * d.H * d.W, d.C, dt))
        _req(sums, "sums", torch.float64)
        nsh = sums.numel() // (2 * d.C)
        if nsh * 2 * d.C != sums.numel():
            raise ValueError("sums must be [nshard,2,C]")
    elif sums is not None:
        raise ValueError("sums without mask_bits")
    extra = (resid.numel() if resid is not None else 0) + (dx.numel() // 16 if mask_bits is not None else 0)
    if bnbwd is not None:
        extra += dy.numel() * (2 if dc_out is not None else 1)  # c read, dc written
    rc = _timed("conv_dgrad", d, dy.element_size(), lambda: lib.msfwsi_panel_dgrad(
        C.byref(d), _p(dy), _p(c), _p(k1), _p(k2), _p(k3), _p(dc_out), _p(wpk), _p(dx), _p(resid), int(resid_stride),
        _p(gapg), float(gap_scale), _p(mask_bits), _p(sums), nsh, _stream()), extra_elems=extra, dtype=dt,
        epi=3 if resid_stride > 1 else 0,
        symbol_override=_panel_symbol(d, dt, d.K, 2 if bnbwd is not None else 0, 3 if resid_stride > 1 else 0,
                                       resid is not None and mask_bits is not None and (gapg is None or d.H * d.W >= 128)))
    if rc == -2:
        return False
    _lib.check(rc, "panel_dgrad")
    return True


def conv_dgrad2(d: ConvDesc, dy, w_cat, dx, src2, bias=None, mask=None, sums=None, src2_pro=None) -> bool:
    """dx = gate(dy . w_cat[0:K] + src2 . w_cat[K:] + bias) in one launch (1x1 only); False if the library has no
    kernel for this shape (the caller then adds the second product as a residual).
    src2_pro = (scale, shift): src2 is a RAW conv output and the operand is relu(scale * src2 + shift), formed inside the
    launch (msfwsi_conv_dgrad2_pro: bit for bit the result on the materialised activation, which is never stored)"""
    lib = _lib.load()
    dt = dy.dtype
    C2 = src2.shape[-1]
    psc = psh = None
    if src2_pro is not None:
        psc, psh = src2_pro
        _req(psc, "pro_scale", torch.float32, C2)
        _req(psh, "pro_shift", torch.float32, C2)
    _req(dy, "dy", dt, d.N * d.P * d.Q * d.K)
    _req(w_cat, "w_cat", dt, (d.K + C2) * d.C)
    _req(dx, "dx", dt, d.N * d.H * d.W * d.C)
    _req(src2, "src2", dt, d.N * d.H * d.W * C2)
    _opt(bias, "bias", torch.float32, d.C)
    mc = msc = msh = None
    nsh = 1
    if mask is not None:
        mc, msc, msh = mask
        _req(mc, "mask_c", dt, d.N * d.H * d.W * d.C)
        _req(msc, "mask_scale", torch.float32, d.C)
        _req(msh, "mask_shift", torch.float32, d.C)
        _req(sums, "sums", torch.float64)
        nsh = sums.numel() // (2 * d.C)
    rc = [0]

    def run():
        if psc is not None:
            rc[0] = lib.msfwsi_conv_dgrad2_pro(C.byref(d), _p(dy), _p(w_cat), _p(dx), _p(src2), int(C2), _p(psc), _p(psh),
                                               _p(bias), _p(mc), _p(msc), _p(msh), _p(sums), nsh, _stream())
        else:
            rc[0] = lib.msfwsi_conv_dgrad2(C.byref(d), _p(dy), _p(w_cat), _p(dx), _p(src2), int(C2), _p(bias), _p(mc), _p(msc),
                                           _p(msh), _p(sums), nsh, _stream())
        if rc[0] != -2:
            _lib.check(rc[0], "conv_dgrad2")

    # algorithmic bytes: with the operand formed in the launch AND the gate read from the same raw tensor, that tensor is one
    # operand (counted once)
    one_tensor = psc is not None and mask is not None and mask[0].data_ptr() == src2.data_ptr()
    _timed("conv_dgrad", d, dy.element_size(), run,
           extra_elems=src2.numel() + (dx.numel() if mask is not None and not one_tensor else 0), dtype=dt, two=True,
           extra_k=int(C2), p2=psc is not None)
    return rc[0] == 0


def conv_wgrad_stationary(d: ConvDesc) -> bool:
    """the output-stationary persistent kernel serves this geometry (64 -> 64, 3x3 / stride 1): its BatchNorm+ReLU
    prologue is free, so callers pass `pro` instead of materialising the activation"""
    return bool(_lib.load().msfwsi_conv_wgrad_stationary(C.byref(d)))


def conv_wgrad(d: ConvDesc, x, dy, dw, pro=None, target_blocks=0):
    lib = _lib.load()
    dt = x.dtype
    _req(x, "x", dt, d.N * d.H * d.W * d.C)
    _req(dy, "dy", dt, d.N * d.P * d.Q * d.K)
    _req(dw, "dw", torch.float32, d.K * d.R * d.S * d.C)
    ps = psh = None
    if pro is not None:
        ps, psh = pro
        _req(ps, "pro_scale", torch.float32, d.C)
        _req(psh, "pro_shift", torch.float32, d.C)
    _timed("conv_wgrad", d, x.element_size(), lambda: _lib.check(
        lib.msfwsi_conv_wgrad(C.byref(d), _p(x), _p(dy), _p(dw), _p(ps), _p(psh), int(target_blocks), _stream()),
        "conv_wgrad"), pro=pro is not None, dtype=x.dtype, same_operand=x.data_ptr() == dy.data_ptr())
    return dw


def conv_wgrad_act(d: ConvDesc, x, dy, dw, pro, act_out, target_blocks=0) -> bool:
    """conv_wgrad of a 1x1 / stride-1 conv with pro = (scale, shift), and act_out <- relu(scale*x + shift) (= bn_act's output)
    from the same register staging; False where the library declines (other geometries, fp32)"""
    lib = _lib.load()
    dt = x.dtype
    _req(x, "x", dt, d.N * d.H * d.W * d.C)
    _req(dy, "dy", dt, d.N * d.P * d.Q * d.K)
    _req(dw, "dw", torch.float32, d.K * d.R * d.S * d.C)
    _req(act_out, "act_out", dt, d.N * d.H * d.W * d.C)
    ps, psh = pro
    _req(ps, "pro_scale", torch.float32, d.C)
    _req(psh, "pro_shift", torch.float32, d.C)
    rc = _timed("conv_wgrad", d, x.element_size(), lambda: lib.msfwsi_conv_wgrad_act(
        C.byref(d), _p(x), _p(dy), _p(dw), _p(ps), _p(psh), _p(act_out), int(target_blocks), _stream()),
        pro=True, dtype=x.dtype, extra_elems=act_out.numel())
    if rc == -2:
        return False
    _lib.check(rc, "conv_wgrad_act")
    return True


def conv_wgrad_store(d: ConvDesc, x, dy, dw):
    """dw = dy^T x, stored (no atomics, dw need not be cleared): one launch per step and tensor -- the heads' Linear layers"""
    lib = _lib.load()
    dt = x.dtype
    _req(x, "x", dt, d.N * d.H * d.W * d.C)
    _req(dy, "dy", dt, d.N * d.P * d.Q * d.K)
    _req(dw, "dw", torch.float32, d.K * d.R * d.S * d.C)
    _timed("conv_wgrad", d, x.element_size(), lambda: _lib.check(
        lib.msfwsi_conv_wgrad_store(C.byref(d), _p(x), _p(dy), _p(dw), _stream()), "conv_wgrad_store"), dtype=dt)
    return dw


def gram(d: ConvDesc, a, A):
    """A[C][1][1][C] (fp32, zeroed by the caller) = a^T a over the pixels of the NHWC activation `a`; the pixel splits
    accumulate in fp64 (msfwsi_gram) so that the BatchNorm statistics the folded tails derive from A do not depend on the
    order of the atomic additions"""
    lib = _lib.load()
    _req(a, "a", a.dtype, d.N * d.H * d.W * d.C)
    _req(A, "A", torch.float32, d.C * d.C)
    A64 = ARENA.zeros((d.C * d.C,), torch.float64, a.device)
    _timed("conv_wgrad", d, a.element_size(), lambda: _lib.check(
        lib.msfwsi_gram(C.byref(d), _p(a), _p(A64), _stream()), "gram"), dtype=a.dtype, same_operand=True)
    add_f64_to_f32(A64, A)
    return A


def stem_wgrad_bnbwd(d: ConvDesc, x, g, c0, k, dw) -> bool:
    """dw += (k1*g + k2*c0 + k3)^T x for the space-to-depth stem: BatchNorm backward applied in the weight-gradient
    kernel's staging (no msfwsi_bn_bwd_apply pass); False if the library has no such kernel for the shape"""
    lib = _lib.load()
    dt = x.dtype
    _req(x, "x", dt, d.N * d.H * d.W * d.C)
    _req(g, "g", dt, d.N * d.P * d.Q * d.K)
    _req(c0, "c0", dt, d.N * d.P * d.Q * d.K)
    _req(dw, "dw", torch.float32, d.K * d.R * d.S * d.C)
    for nm, t in zip(("k1", "k2", "k3"), k):
        _req(t, nm, torch.float32, d.K)
    rc = [0]

    def run():
        rc[0] = lib.msfwsi_stem_wgrad_bnbwd(C.byref(d), _p(x), _p(g), _p(c0), _p(k[0]), _p(k[1]), _p(k[2]), _p(dw),
                                            _stream())
        if rc[0] != -2:
            _lib.check(rc[0], "stem_wgrad_bnbwd")

    tc = "DF16_" if dt == torch.float16 else _TCODE[x.element_size()]
    _timed("conv_wgrad", d, x.element_size(), run, extra_elems=c0.numel(), dtype=dt,
           symbol_override=f"stem_wgrad_os_kernelI{tc}Lb1EE")
    return rc[0] == 0


# ------------------------------------------------------------------------------------------------
def bn_finalize(sums, count, gamma, beta, eps, momentum, running_mean, running_var, nbt, scale, shift, mean,
                invstd):
    lib = _lib.load()
    _req(sums, "sums", torch.float64)
    Cn = scale.numel()
    nsh = sums.numel() // (2 * Cn)
    if nsh * 2 * Cn != sums.numel():
        raise ValueError("sums must be [nshard,2,C]")
    for n, t in (("gamma", gamma), ("beta", beta), ("running_mean", running_mean), ("running_var", running_var)):
        _opt(t, n, torch.float32, Cn)
    _opt(nbt, "num_batches_tracked", torch.int64, 1)
    for n, t in (("scale", scale), ("shift", shift), ("mean", mean), ("invstd", invstd)):
        _req(t, n, torch.float32, Cn)
    _lib.check(lib.msfwsi_bn_finalize(_p(sums), nsh, Cn, float(count), _p(gamma), _p(beta), float(eps),
                                      float(momentum), _p(running_mean), _p(running_var), _p(nbt), _p(scale),
                                      _p(shift), _p(mean), _p(invstd), _stream()), "bn_finalize")


def bn_eval_coeffs(running_mean, running_var, gamma, beta, eps, scale, shift, mean, invstd):
    """eval-mode BatchNorm: the affine map from the running statistics (no reduction, no exchange)"""
    lib = _lib.load()
    Cn = scale.numel()
    for n, t in (("running_mean", running_mean), ("running_var", running_var), ("scale", scale), ("shift", shift),
                 ("mean", mean), ("invstd", invstd)):
        _req(t, n, torch.float32, Cn)
    _opt(gamma, "gamma", torch.float32, Cn)
    _opt(beta, "beta", torch.float32, Cn)
    _lib.check(lib.msfwsi_bn_eval_coeffs(_p(running_mean), _p(running_var), _p(gamma), _p(beta), float(eps), Cn,
                                         _p(scale), _p(shift), _p(mean), _p(invstd), _stream()), "bn_eval_coeffs")


def shard_sum(sums, out):
    lib = _lib.load()
    _req(sums, "sums", torch.float64)
    _req(out, "out", torch.float64)
    n = out.numel()
    nsh = sums.numel() // n
    if nsh * n != sums.numel():
        raise ValueError("shard_sum: size mismatch")
    _lib.check(lib.msfwsi_shard_sum(_p(sums), nsh, n, _p(out), _stream()), "shard_sum")
    return out


def bn_act(c, scale, shift, out, ident=None, id_scale=None, id_shift=None, relu=True):
    lib = _lib.load()
    Cn = scale.numel()
    M = c.numel() // Cn
    _req(c, "c", None, M * Cn)
    _req(out, "out", c.dtype, M * Cn)
    _req(scale, "scale", torch.float32, Cn)
    _req(shift, "shift", torch.float32, Cn)
    _opt(ident, "ident", c.dtype, M * Cn)
    _opt(id_scale, "id_scale", torch.float32, Cn)
    _opt(id_shift, "id_shift", torch.float32, Cn)
    _stream_timed("bn_act", c.element_size() * M * Cn * (2 + (ident is not None)), lambda: _lib.check(
        lib.msfwsi_bn_act(dt_of(c), _p(c), _p(scale), _p(shift), _p(ident), _p(id_scale), _p(id_shift),
                          int(bool(relu)), _p(out), M, Cn, _stream()), "bn_act"))
    return out


def bn_act_sum(c, scale, shift, out, sums):
    """out = relu(scale*c+shift); sums[C] (fp64) = column sums of out"""
    lib = _lib.load()
    Cn = scale.numel()
    M = c.numel() // Cn
    _req(c, "c", None, M * Cn)
    _req(out, "out", c.dtype, M * Cn)
    _req(scale, "scale", torch.float32, Cn)
    _req(shift, "shift", torch.float32, Cn)
    _req(sums, "sums", torch.float64, Cn)
    # sharded accumulation (one replica would serialise ~2000 workgroups on C addresses), then one tiny reduction
    part = ARENA.zeros((NSHARD, 1, Cn), torch.float64, c.device)
    _stream_timed("bn_act_sum", c.element_size() * M * Cn * 2, lambda: _lib.check(
        lib.msfwsi_bn_act_sum(dt_of(c), _p(c), _p(scale), _p(shift), _p(out), _p(part), NSHARD, M, Cn, _stream()),
        "bn_act_sum"))
    shard_sum(part, sums)
    return out


def block_end_bwd(dy, y, gapg, gap_scale, c_main, c_ds, g, sums, HW):
    lib = _lib.load()
    Cn = y.shape[-1]
    M = y.numel() // Cn
    _req(y, "y")
    _opt(dy, "dy", y.dtype, M * Cn)
    _opt(gapg, "gapg", y.dtype, (M // HW) * Cn)
    _opt(c_main, "c_main", y.dtype, M * Cn)
    _opt(c_ds, "c_ds", y.dtype, M * Cn)
    _req(g, "g", y.dtype, M * Cn)
    _req(sums, "sums", torch.float64)
    nsh = sums.numel() // (3 * Cn)
    if nsh * 3 * Cn != sums.numel() or M % HW != 0:
        raise ValueError("block_end_bwd: bad sums / HW")
    _stream_timed("block_end_bwd", y.element_size() * M * Cn * (3 + (c_main is not None) + (c_ds is not None)),
                  lambda: _lib.check(lib.msfwsi_block_end_bwd(dt_of(y), _p(dy), _p(y), _p(gapg), float(gap_scale),
                                                              _p(c_main), _p(c_ds), _p(g), _p(sums), nsh, M, HW, Cn,
                                                              _stream()), "block_end_bwd"))


def act_bwd_reduce(da, c, scale, shift, g, sums):
    lib = _lib.load()
    Cn = c.shape[-1]
    M = c.numel() // Cn
    _req(da, "da", c.dtype, M * Cn)
    _req(c, "c")
    _opt(scale, "scale", torch.float32, Cn)
    _opt(shift, "shift", torch.float32, Cn)
    _opt(g, "g", c.dtype, M * Cn)
    _req(sums, "sums", torch.float64)
    nsh = sums.numel() // (2 * Cn)
    if nsh * 2 * Cn != sums.numel():
        raise ValueError("act_bwd_reduce: sums must be [nshard,2,C]")
    _stream_timed("act_bwd_reduce", c.element_size() * M * Cn * (2 + (g is not None)), lambda: _lib.check(
        lib.msfwsi_act_bwd_reduce(dt_of(c), _p(da), _p(c), _p(scale), _p(shift), _p(g), _p(sums), nsh, M, Cn, _stream()),
        "act_bwd_reduce"))


def bn_bwd_finalize(sums, nslots, which, count, gamma, mean, invstd, dgamma, dbeta, k1, k2, k3):
    lib = _lib.load()
    Cn = mean.numel()
    _req(sums, "sums", torch.float64)
    nsh = sums.numel() // (nslots * Cn)
    if nsh * nslots * Cn != sums.numel():
        raise ValueError("bn_bwd_finalize: sums size")
    for n, t in (("gamma", gamma), ("dgamma", dgamma), ("dbeta", dbeta)):
        _opt(t, n, torch.float32, Cn)
    for n, t in (("mean", mean), ("invstd", invstd), ("k1", k1), ("k2", k2), ("k3", k3)):
        _req(t, n, torch.float32, Cn)
    _lib.check(lib.msfwsi_bn_bwd_finalize(_p(sums), nsh, nslots, which, Cn, float(count), _p(gamma), _p(mean),
                                          _p(invstd), _p(dgamma), _p(dbeta), _p(k1), _p(k2), _p(k3), _stream()),
               "bn_bwd_finalize")


def bn_bwd_apply(g, c, k1, k2, k3, dc):
    lib = _lib.load()
    Cn = k1.numel()
    M = c.numel() // Cn
    _req(g, "g", c.dtype, M * Cn)
    _req(c, "c")
    _req(dc, "dc", c.dtype, M * Cn)
    for n, t in (("k1", k1), ("k2", k2), ("k3", k3)):
        _req(t, n, torch.float32, Cn)
    _stream_timed("bn_bwd_apply", c.element_size() * M * Cn * 3, lambda: _lib.check(
        lib.msfwsi_bn_bwd_apply(dt_of(c), _p(g), _p(c), _p(k1), _p(k2), _p(k3), _p(dc), M, Cn, _stream()),
        "bn_bwd_apply"))
    return dc


# ------------------------------------------------------------------------------------------------
def nchw_to_nhwc(x, y, CP):
    lib = _lib.load()
    N, Cc, H, W = x.shape
    _req(x, "x", torch.float32)
    _req(y, "y", None, N * H * W * CP)
    _stream_timed("nchw_to_nhwc", 4 * x.numel() + y.element_size() * y.numel(), lambda: _lib.check(
        lib.msfwsi_nchw_to_nhwc(dt_of(y), _p(x), _p(y), N, Cc, H, W, CP, _stream()), "nchw_to_nhwc"))
    return y


def nchw_to_s2d(x, y):
    """fp32 NCHW [N,3,H,W] (H, W even) -> storage [N,H/2,W/2,16]: channel (a*2+b)*3+c = x[c][2i+a][2j+b]"""
    lib = _lib.load()
    N, Cc, H, W = x.shape
    if Cc != 3 or H % 2 or W % 2:
        raise ValueError("nchw_to_s2d: [N,3,H,W] with even H, W")
    _req(x, "x", torch.float32)
    _req(y, "y", None, N * (H // 2) * (W // 2) * 16)
    _stream_timed("nchw_to_nhwc", 4 * x.numel() + y.element_size() * y.numel(), lambda: _lib.check(
        lib.msfwsi_nchw_to_s2d(dt_of(y), _p(x), _p(y), N, H, W, _stream()), "nchw_to_s2d"))
    return y


def stem_s2d_weights(w, out):
    """fp32 [K][7][7][3] -> storage [K][4][4][16] (the 4x4 / stride-1 form of the 7x7 / stride-2 stem)"""
    lib = _lib.load()
    K = w.shape[0]
    _req(w, "w", torch.float32, K * 147)
    _req(out, "out", None, K * 256)
    _lib.check(lib.msfwsi_stem_s2d_weights(dt_of(out), _p(w), _p(out), K, _stream()), "stem_s2d_weights")
    return out


def stem_s2d_wfold(dw2, dw):
    """dw [K][7][7][3] += gather of dw2 [K][4][4][16]"""
    lib = _lib.load()
    K = dw.shape[0]
    _req(dw2, "dw2", torch.float32, K * 256)
    _req(dw, "dw", torch.float32, K * 147)
    _lib.check(lib.msfwsi_stem_s2d_wfold(_p(dw2), _p(dw), K, _stream()), "stem_s2d_wfold")


def stem_pool_fwd(c0, scale, shift, out, argmax, N, H, W, Cn):
    lib = _lib.load()
    P, Q = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    _req(c0, "c0", None, N * H * W * Cn)
    _req(out, "out", c0.dtype, N * P * Q * Cn)
    _req(argmax, "argmax", torch.uint8, N * P * Q * Cn)
    _req(scale, "scale", torch.float32, Cn)
    _req(shift, "shift", torch.float32, Cn)
    _stream_timed("stem_pool_fwd", c0.element_size() * (N * H * W * Cn + N * P * Q * Cn) + N * P * Q * Cn,
                  lambda: _lib.check(lib.msfwsi_stem_pool_fwd(dt_of(c0), _p(c0), _p(scale), _p(shift), _p(out),
                                                              _p(argmax), N, H, W, Cn, _stream()), "stem_pool_fwd"))


def stem_pool_bwd(dp, argmax, c0, scale, shift, g0, sums, N, H, W, Cn, k=None, dact=None):
    """k=None: sums += {sum g, sum g*c0}, g0 (optional) = g;  k=(k1,k2,k3): g0 = k1*g + k2*c0 + k3;
    dact: gradient of the stem activation itself (U-Net skip), added before the gate"""
    lib = _lib.load()
    P, Q = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    _req(c0, "c0", None, N * H * W * Cn)
    _req(dp, "dp", c0.dtype, N * P * Q * Cn)
    _req(argmax, "argmax", torch.uint8, N * P * Q * Cn)
    _opt(g0, "g0", c0.dtype, N * H * W * Cn)
    _req(scale, "scale", torch.float32, Cn)
    _req(shift, "shift", torch.float32, Cn)
    nsh = 1
    if sums is not None:
        _req(sums, "sums", torch.float64)
        nsh = sums.numel() // (2 * Cn)
        if nsh * 2 * Cn != sums.numel():
            raise ValueError("stem_pool_bwd: sums must be [nshard,2,C]")
    _opt(dact, "dact", c0.dtype, N * H * W * Cn)
    k1 = k2 = k3 = None
    if k is not None:
        k1, k2, k3 = k
        for nm, t in (("k1", k1), ("k2", k2), ("k3", k3)):
            _req(t, nm, torch.float32, Cn)
    _stream_timed("stem_pool_bwd", c0.element_size() * (N * H * W * Cn * (1 + (g0 is not None)) + N * P * Q * Cn)
                  + N * P * Q * Cn, lambda: _lib.check(
        lib.msfwsi_stem_pool_bwd(dt_of(c0), _p(dp), _p(argmax), _p(c0), _p(scale), _p(shift), _p(g0), _p(sums), nsh,
                                 _p(k1), _p(k2), _p(k3), _p(dact), N, H, W, Cn, _stream()), "stem_pool_bwd"))


def gap_fwd(y, out, N, HW, Cn):
    lib = _lib.load()
    _req(y, "y", None, N * HW * Cn)
    _req(out, "out", y.dtype, N * Cn)
    _stream_timed("gap_fwd", y.element_size() * N * HW * Cn, lambda: _lib.check(
        lib.msfwsi_gap_fwd(dt_of(y), _p(y), _p(out), N, HW, Cn, _stream()), "gap_fwd"))
    return out


def gap_fwd_stride2(y, out, sout, N, H, W, Cn):
    """gap_fwd + pixel_stride(stride 2) in one pass: out [N][C] pooled features, sout = y[:, ::2, ::2, :]"""
    lib = _lib.load()
    _req(y, "y", None, N * H * W * Cn)
    _req(out, "out", y.dtype, N * Cn)
    _req(sout, "sout", y.dtype, N * ((H + 1) // 2) * ((W + 1) // 2) * Cn)
    _stream_timed("gap_fwd", y.element_size() * (N * H * W * Cn + sout.numel()), lambda: _lib.check(
        lib.msfwsi_gap_fwd_stride2(dt_of(y), _p(y), _p(out), _p(sout), N, H, W, Cn, _stream()), "gap_fwd_stride2"))
    return out


def fold_dots(W, Mm, out):
    """out[k] = sum_c W[k][c] * Mm[k][c]   (fp64; the sum over pixels of g*c for c = W a, Mm = g^T a)"""
    lib = _lib.load()
    K = out.numel()
    Cn = W.numel() // K
    _req(W, "W", torch.float32, K * Cn)
    _req(Mm, "M", torch.float32, K * Cn)
    _req(out, "out", torch.float64, K)
    _lib.check(lib.msfwsi_fold_dots(_p(W), _p(Mm), _p(out), K, Cn, _stream()), "fold_dots")


def fold_weights(W, Mm, WA, k1, k2, k3, sa, dW, Wk1, Wk2, bvec):
    """dW += k1 o Mm + k2 o WA + k3 (x) sa;  Wk1 = k1 o W;  Wk2 = k2 o W;  bvec (fp32, zeroed by the caller) += W^T k3,
    accumulated in fp64 (the order of the row blocks' atomic additions then does not show in the fp32 result)"""
    lib = _lib.load()
    K = k1.numel()
    Cn = W.numel() // K
    for nm, t in (("W", W), ("M", Mm), ("WA", WA), ("dW", dW), ("Wk1", Wk1), ("Wk2", Wk2)):
        _req(t, nm, torch.float32, K * Cn)
    for nm, t in (("k1", k1), ("k2", k2), ("k3", k3)):
        _req(t, nm, torch.float32, K)
    _req(sa, "sa", torch.float64, Cn)
    _req(bvec, "bvec", torch.float32, Cn)
    b64 = ARENA.zeros((Cn,), torch.float64, bvec.device)
    _lib.check(lib.msfwsi_fold_weights(_p(W), _p(Mm), _p(WA), _p(k1), _p(k2), _p(k3), _p(sa), _p(dW), _p(Wk1), _p(Wk2),
                                       _p(b64), K, Cn, _stream()), "fold_weights")
    add_f64_to_f32(b64, bvec)


def colsum(x, sums):
    lib = _lib.load()
    Cn = sums.numel()
    M = x.numel() // Cn
    _req(x, "x", None, M * Cn)
    _req(sums, "sums", torch.float64, Cn)
    if M < 4096:  # few workgroups: straight into the result
        _lib.check(lib.msfwsi_colsum(dt_of(x), _p(x), _p(sums), 1, M, Cn, _stream()), "colsum")
        return
    part = ARENA.zeros((NSHARD, 1, Cn), torch.float64, x.device)
    _lib.check(lib.msfwsi_colsum(dt_of(x), _p(x), _p(part), NSHARD, M, Cn, _stream()), "colsum")
    tot = torch.empty(Cn, dtype=torch.float64, device=x.device)
    shard_sum(part, tot)
    _lib.check(lib.msfwsi_add_f64(_p(tot), _p(sums), Cn, _stream()), "add_f64")


def colstats(x, stats):
    """stats[nshard][2][C] += {column sums, column sums of squares} of x [M][C], fp64 from the first add on"""
    lib = _lib.load()
    _req(stats, "stats", torch.float64)
    nsh = stats.shape[0]
    Cn = stats.shape[-1]
    M = x.numel() // Cn
    _req(x, "x", None, M * Cn)
    if stats.numel() != nsh * 2 * Cn:
        raise ValueError("stats must be [nshard,2,C]")
    _lib.check(lib.msfwsi_colstats(dt_of(x), _p(x), _p(stats), nsh, M, Cn, _stream()), "colstats")


def add_f64_to_f32(src, dst, alpha=1.0):
    lib = _lib.load()
    _req(src, "src", torch.float64)
    _req(dst, "dst", torch.float32, src.numel())
    _lib.check(lib.msfwsi_add_f64_to_f32(_p(src), _p(dst), src.numel(), float(alpha), _stream()), "add_f64_to_f32")


def rows_permute(inp, idx, out, B, K, Cn, scatter=False, accumulate=False):
    lib = _lib.load()
    _req(inp, "in", None, B * K * Cn)
    _req(out, "out", inp.dtype, B * K * Cn)
    _req(idx, "idx", torch.int64, B * K)
    _lib.check(lib.msfwsi_rows_permute(dt_of(inp), _p(inp), _p(idx), _p(out), B, K, Cn, int(scatter),
                                       int(accumulate), _stream()), "rows_permute")
    return out


def copy2d(src, src_off, src_ld, dst, dst_off, dst_ld, rows, cols, accumulate=False):
    """dst.flat[dst_off + r*dst_ld + c] (+)= src.flat[src_off + r*src_ld + c]"""
    lib = _lib.load()
    _req(src, "src")
    _req(dst, "dst", src.dtype)
    if src_off + (rows - 1) * src_ld + cols > src.numel() or dst_off + (rows - 1) * dst_ld + cols > dst.numel():
        raise ValueError("copy2d: out of range")
    es = src.element_size()
    v = vec_of(src.dtype)
    if src_off % v or dst_off % v:
        raise ValueError("copy2d: offsets must be 16-byte aligned")
    _lib.check(lib.msfwsi_copy2d(dt_of(src), src.data_ptr() + src_off * es, src_ld, dst.data_ptr() + dst_off * es,
                                 dst_ld, rows, cols, int(accumulate), _stream()), "copy2d")


# ------------------------------------------------------------------------------------------------
def cosine_loss(p, z, coef, loss_accum, dp=None, loss_scale=None, eps=1e-8):
    lib = _lib.load()
    rows, dd = p.shape
    _req(p, "p")
    _req(z, "z", p.dtype, rows * dd)
    _opt(dp, "dp", p.dtype, rows * dd)
    _opt(loss_accum, "loss_accum", torch.float64, 1)
    _opt(loss_scale, "loss_scale", torch.float32, 1)
    _lib.check(lib.msfwsi_cosine_loss(dt_of(p), _p(p), _p(z), rows, dd, float(coef), _p(loss_scale), float(eps),
                                      _p(loss_accum), _p(dp), _stream()), "cosine_loss")


def row_l2norm(x, out, inv, eps=1e-12):
    """out = x / max(||x||_2, eps) per row (F.normalize), inv[row] = the factor"""
    lib = _lib.load()
    rows, dd = x.shape
    _req(x, "x")
    _req(out, "out", x.dtype, rows * dd)
    _req(inv, "inv", torch.float32, rows)
    _lib.check(lib.msfwsi_row_l2norm(dt_of(x), _p(x), _p(out), _p(inv), rows, dd, float(eps), _stream()), "row_l2norm")
    return out


def row_l2norm_bwd(xhat, dxhat, inv, dx):
    lib = _lib.load()
    rows, dd = xhat.shape
    _req(xhat, "xhat")
    _req(dxhat, "dxhat", xhat.dtype, rows * dd)
    _req(dx, "dx", xhat.dtype, rows * dd)
    _req(inv, "inv", torch.float32, rows)
    _lib.check(lib.msfwsi_row_l2norm_bwd(dt_of(xhat), _p(xhat), _p(dxhat), _p(inv), _p(dx), rows, dd, _stream()),
               "row_l2norm_bwd")
    return dx


def softmax_ce(logits, label0: int, inv_tau: float, coef: float, loss_accum, grad_scale=None, write_grad=True):
    """cross entropy of logits [rows, n] / tau against label = label0 + row; in place: logits <- d loss / d logits"""
    lib = _lib.load()
    rows, n = logits.shape
    _req(logits, "logits")
    _opt(loss_accum, "loss_accum", torch.float64, 1)
    _opt(grad_scale, "grad_scale", torch.float32, 1)
    _lib.check(lib.msfwsi_softmax_ce(dt_of(logits), _p(logits), rows, n, int(label0), float(inv_tau), float(coef),
                                     _p(grad_scale), _p(loss_accum), int(bool(write_grad)), _stream()), "softmax_ce")


def nonfinite_check(g, found):
    lib = _lib.load()
    _req(g, "g", torch.float32)
    _req(found, "found", torch.float32, 1)
    _lib.check(lib.msfwsi_nonfinite_check(_p(g), g.numel(), _p(found), _stream()), "nonfinite_check")


def scaler_update(scale, tracker, found, growth_factor, backoff_factor, growth_interval):
    lib = _lib.load()
    _req(scale, "scale", torch.float32, 1)
    _req(tracker, "tracker", torch.int32, 1)
    _req(found, "found", torch.float32, 1)
    _lib.check(lib.msfwsi_scaler_update(_p(scale), _p(tracker), _p(found), float(growth_factor),
                                        float(backoff_factor), int(growth_interval), _stream()), "scaler_update")


def adam(p, g, m, v, lr, beta1, beta2, eps, step, loss_scale=None, found=None, p_lowp=None):
    """p_lowp: bf16 / fp16 compute copy of p refreshed in the same pass; step: host int, or a device int32[1]
    tensor (advanced by adam_step_advance) from which the kernel forms the bias corrections itself"""
    lib = _lib.load()
    n = p.numel()
    for nm, t in (("p", p), ("g", g), ("m", m), ("v", v)):
        _req(t, nm, torch.float32, n)
    _opt(loss_scale, "loss_scale", torch.float32, 1)
    _opt(found, "found", torch.float32, 1)
    code = 0
    if p_lowp is not None:
        _req(p_lowp, "p_lowp", None, n)
        code = LOWP[p_lowp.dtype]
    step_dev = None
    if isinstance(step, torch.Tensor):
        _req(step, "step", torch.int32, 1)
        step_dev, step = step, 0
    _stream_timed("adam", n * (28 + (2 if p_lowp is not None else 0)), lambda: _lib.check(
        lib.msfwsi_adam(_p(p), _p(g), _p(m), _p(v), n, float(lr), float(beta1), float(beta2), float(eps), int(step),
                        _p(step_dev), _p(loss_scale), _p(found), _p(p_lowp), code, _stream()), "adam"))


def adam_step_advance(step, found=None):
    """step[0] += 1 unless found[0] > 0 (a step the GradScaler skips does not advance Adam's step count)"""
    lib = _lib.load()
    _req(step, "step", torch.int32, 1)
    _opt(found, "found", torch.float32, 1)
    _lib.check(lib.msfwsi_adam_step_advance(_p(step), _p(found), _stream()), "adam_step_advance")


def cast_lowp(src, dst):
    """fp32 -> bf16 / fp16 (dst.dtype)"""
    lib = _lib.load()
    _req(src, "src", torch.float32)
    _req(dst, "dst", None, src.numel())
    _lib.check(lib.msfwsi_cast_lowp(LOWP[dst.dtype], _p(src), _p(dst), src.numel(), _stream()), "cast_lowp")
    return dst


def upcast_f32(src, dst=None):
    """bf16 / fp16 -> fp32"""
    lib = _lib.load()
    _req(src, "src")
    if dst is None:
        dst = torch.empty(src.shape, dtype=torch.float32, device=src.device)
    _req(dst, "dst", torch.float32, src.numel())
    _lib.check(lib.msfwsi_upcast_f32(LOWP[src.dtype], _p(src), _p(dst), src.numel(), _stream()), "upcast_f32")
    return dst


def zero_slot(sums, slot: int):
    """sums[:, slot, :] = 0 for a sharded fp64 accumulator [nshard][slots][C]"""
    lib = _lib.load()
    _req(sums, "sums", torch.float64)
    nsh, nslots, Cn = sums.shape
    _lib.check(lib.msfwsi_zero_f64_2d(sums.data_ptr() + slot * Cn * 8, nsh, Cn, nslots * Cn, _stream()), "zero_slot")


def pad_cast(src, dst, rows, Cn, CP):
    lib = _lib.load()
    _req(src, "src", torch.float32, rows * Cn)
    _req(dst, "dst", None, rows * CP)
    _lib.check(lib.msfwsi_pad_cast(dt_of(dst), _p(src), _p(dst), rows, Cn, CP, _stream()), "pad_cast")
    return dst


def unpad_add(src, dst, rows, Cn, CP):
    lib = _lib.load()
    _req(src, "src", torch.float32, rows * CP)
    _req(dst, "dst", torch.float32, rows * Cn)
    _lib.check(lib.msfwsi_unpad_add(_p(src), _p(dst), rows, Cn, CP, _stream()), "unpad_add")


# ------------------------------------------------------------------------------------------------
# 3x3 / stride-1 halo kernel (csrc/conv3x3.hip)
# ------------------------------------------------------------------------------------------------
def conv3x3_supported(d: ConvDesc) -> bool:
    return bool(_lib.load().msfwsi_conv3x3_supported(C.byref(d)))


def conv3x3_stationary(d: ConvDesc) -> bool:
    """the weights-stationary persistent kernel serves this geometry (64 -> 64 channels, 2-byte types)"""
    return bool(_lib.load().msfwsi_conv3x3_stationary(C.byref(d)))


def conv3x3_fwd(d: ConvDesc, x, w, y, stats=None, pro=None):
    """pro = (scale, shift) of the producer BatchNorm: x is its raw conv output (only where conv3x3_stationary(d))"""
    lib = _lib.load()
    dt = x.dtype
    ps = psh = None
    if pro is not None:
        ps, psh = pro
        _req(ps, "pro_scale", torch.float32, d.C)
        _req(psh, "pro_shift", torch.float32, d.C)
    _req(x, "x", dt, d.N * d.H * d.W * d.C)
    _req(w, "w", dt, d.K * 9 * d.C)
    _req(y, "y", dt, d.N * d.P * d.Q * d.K)
    nsh = 1
    if stats is not None:
        _req(stats, "stats", torch.float64)
        nsh = stats.shape[0]
        if stats.numel() != nsh * 2 * d.K:
            raise ValueError("stats must be [nshard,2,K]")
    _timed("conv_fwd", d, x.element_size(), lambda: _lib.check(
        lib.msfwsi_conv3x3_fwd(C.byref(d), _p(x), _p(w), _p(y), _p(stats), nsh, _p(ps), _p(psh), _stream()),
        "conv3x3_fwd"), halo=True, dtype=x.dtype)
    return y


def conv3x3_dgrad(d: ConvDesc, dy, w, dx, resid=None, mask=None, sums=None):
    lib = _lib.load()
    dt = dy.dtype
    _req(dy, "dy", dt, d.N * d.P * d.Q * d.K)
    _req(w, "w", dt, d.K * 9 * d.C)
    _req(dx, "dx", dt, d.N * d.H * d.W * d.C)
    _opt(resid, "resid", dt, d.N * d.H * d.W * d.C)
    mc = msc = msh = None
    nsh = 1
    if mask is not None:
        mc, msc, msh = mask
        _req(mc, "mask_c", dt, d.N * d.H * d.W * d.C)
        _req(msc, "mask_scale", torch.float32, d.C)
        _req(msh, "mask_shift", torch.float32, d.C)
        _req(sums, "sums", torch.float64)
        nsh = sums.numel() // (2 * d.C)
        if nsh * 2 * d.C != sums.numel():
            raise ValueError("sums must be [nshard,2,C]")
    elif sums is not None:
        raise ValueError("sums without mask")
    _timed("conv_dgrad", d, dy.element_size(), lambda: _lib.check(
        lib.msfwsi_conv3x3_dgrad(C.byref(d), _p(dy), _p(w), _p(dx), _p(resid), _p(mc), _p(msc), _p(msh), _p(sums),
                                 nsh, _stream()), "conv3x3_dgrad"),
        extra_elems=(dx.numel() if resid is not None else 0) + (dx.numel() if mask is not None else 0), halo=True,
        dtype=dy.dtype)
    return dx


# ------------------------------------------------------------------------------------------------
# image-stationary 3x3 of the deep layers (csrc/img3x3.hip)
# ------------------------------------------------------------------------------------------------
def img3x3_supported(d: ConvDesc) -> bool:
    """conv2 of layer2 / layer3 (src/models/resnet.py:128 at 28x28x128 / 14x14x256, 16-bit storage)"""
    return bool(_lib.load().msfwsi_img3x3_supported(C.byref(d)))


def img3x3_pack_weights(w, wpk, dgrad):
    """wpk <- w [K][3][3][C] in the fragment order the kernel streams (dgrad True / 1: transposed, taps flipped; 2: the
    pass order of the strided gradient, img3x3_s2_dgrad)"""
    K, Cc = w.shape[0], w.shape[-1]
    _req(w, "w", w.dtype, K * 9 * Cc)
    _req(wpk, "wpk", w.dtype, K * 9 * Cc)
    _lib.check(_lib.load().msfwsi_img3x3_pack_weights(dt_of(w), _p(w), _p(wpk), K, Cc, int(dgrad), _stream()),
               "img3x3_pack_weights")
    return wpk


def img3x3_s2_dgrad_supported(d: ConvDesc) -> bool:
    """the strided conv2 of layer2.0 / layer3.0 (resnet.py:128 with stride 2): dy 28x28x128 / 14x14x256, 16-bit storage"""
    return bool(_lib.load().msfwsi_img3x3_s2_dgrad_supported(C.byref(d)))


def img3x3_s2_dgrad(d: ConvDesc, dy, wpk, dx, bnbwd=None, dc_out=None, mask=None, sums=None, act_out=None) -> bool:
    """input gradient of a 3x3 / stride 2 conv in ONE launch (d: the forward conv); arguments as img3x3_dgrad, bnbwd / dc_out
    at dy's resolution, mask / act_out / sums at dx's.  False where the geometry is not served."""
    lib = _lib.load()
    dt = dy.dtype
    n_out = d.N * d.P * d.Q * d.K
    n_in = d.N * d.H * d.W * d.C
    _req(dy, "dy", dt, n_out)
    _req(wpk, "wpk", dt, d.K * 9 * d.C)
    _req(dx, "dx", dt, n_in)
    cc = k1 = k2 = k3 = None
    if bnbwd is not None:
        cc, k1, k2, k3 = bnbwd
        _req(cc, "c", dt, n_out)
        for nm, t in (("k1", k1), ("k2", k2), ("k3", k3)):
            _req(t, nm, torch.float32, d.K)
    _opt(dc_out, "dc_out", dt, n_out)
    _opt(act_out, "act_out", dt, n_in)
    mc = msc = msh = None
    nsh = 1
    if mask is not None:
        mc, msc, msh = mask
        _req(mc, "mask_c", dt, n_in)
        _req(msc, "mask_scale", torch.float32, d.C)
        _req(msh, "mask_shift", torch.float32, d.C)
        _req(sums, "sums", torch.float64)
        nsh = sums.numel() // (2 * d.C)
        if nsh * 2 * d.C != sums.numel():
            raise ValueError("sums must be [nshard,2,C]")
    elif sums is not None:
        raise ValueError("sums without mask")
    bh = {28: 7, 14: 14}.get(d.P, 0)
    t = "DF16_" if dt == torch.float16 else "DF16b"
    rc = _timed("conv_dgrad", d, dy.element_size(), lambda: lib.msfwsi_img3x3_s2_dgrad(
        C.byref(d), _p(dy), _p(cc), _p(k1), _p(k2), _p(k3), _p(dc_out), _p(wpk), _p(dx), _p(mc), _p(msc), _p(msh), _p(act_out),
        _p(sums), nsh, _stream()),
        extra_elems=n_in * ((mask is not None) + (act_out is not None)) + n_out * ((bnbwd is not None) + (dc_out is not None)),
        dtype=dy.dtype, symbol_override=f"img3x3_s2d_kernelI{t}Li{d.C}ELi{bh}ELi{d.Q}ELi{2 if bnbwd is not None else 0}EE")
    if rc == -2:
        return False
    _lib.check(rc, "img3x3_s2_dgrad")
    return True


def _img3_symbol(d: ConvDesc, dt, pro: int, dgrad: bool) -> str:
    t = "DF16_" if dt == torch.float16 else "DF16b"
    bh = {14: 14, 28: 7, 56: 4}.get(d.H, 0)
    return f"img3x3_kernelI{t}Li{d.C}ELi{d.K}ELi{bh}ELi{d.W}ELi{pro}ELb{int(dgrad)}ELi1EE"  # (TN = 1: csrc/img3x3.hip)


def img3x3_fwd(d: ConvDesc, x, wpk, y, stats=None, pro=None) -> bool:
    """False (nothing launched) where the geometry is not served; pro = (scale, shift): x is the producer's raw output"""
    lib = _lib.load()
    dt = x.dtype
    ps = psh = None
    if pro is not None:
        ps, psh = pro
        _req(ps, "pro_scale", torch.float32, d.C)
        _req(psh, "pro_shift", torch.float32, d.C)
    _req(x, "x", dt, d.N * d.H * d.W * d.C)
    _req(wpk, "wpk", dt, d.K * 9 * d.C)
    _req(y, "y", dt, d.N * d.P * d.Q * d.K)
    nsh = 1
    if stats is not None:
        _req(stats, "stats", torch.float64)
        nsh = stats.shape[0]
        if stats.numel() != nsh * 2 * d.K:
            raise ValueError("stats must be [nshard,2,K]")
    rc = _timed("conv_fwd", d, x.element_size(), lambda: lib.msfwsi_img3x3_fwd(
        C.byref(d), _p(x), _p(ps), _p(psh), _p(wpk), _p(y), _p(stats), nsh, _stream()), halo=True, dtype=x.dtype,
        symbol_override=_img3_symbol(d, dt, 1 if pro is not None else 0, False))
    if rc == -2:
        return False
    _lib.check(rc, "img3x3_fwd")
    return True


def img3x3_dgrad(d: ConvDesc, dy, wpk, dx, bnbwd=None, dc_out=None, mask=None, sums=None, act_out=None) -> bool:
    """bnbwd = (c, k1, k2, k3): the gradient operand is k1*dy + k2*c + k3 (BatchNorm backward fused into the staging),
    written to dc_out when given; mask / sums as conv3x3_dgrad; act_out receives relu(mask_scale*mask_c + mask_shift) (=
    bn_act of the gating layer).  False where the geometry is not served."""
    lib = _lib.load()
    dt = dy.dtype
    n_out = d.N * d.P * d.Q * d.K
    _req(dy, "dy", dt, n_out)
    _req(wpk, "wpk", dt, d.K * 9 * d.C)
    _req(dx, "dx", dt, d.N * d.H * d.W * d.C)
    cc = k1 = k2 = k3 = None
    if bnbwd is not None:
        cc, k1, k2, k3 = bnbwd
        _req(cc, "c", dt, n_out)
        for nm, t in (("k1", k1), ("k2", k2), ("k3", k3)):
            _req(t, nm, torch.float32, d.K)
    _opt(dc_out, "dc_out", dt, n_out)
    _opt(act_out, "act_out", dt, d.N * d.H * d.W * d.C)
    mc = msc = msh = None
    nsh = 1
    if mask is not None:
        mc, msc, msh = mask
        _req(mc, "mask_c", dt, d.N * d.H * d.W * d.C)
        _req(msc, "mask_scale", torch.float32, d.C)
        _req(msh, "mask_shift", torch.float32, d.C)
        _req(sums, "sums", torch.float64)
        nsh = sums.numel() // (2 * d.C)
        if nsh * 2 * d.C != sums.numel():
            raise ValueError("sums must be [nshard,2,C]")
    elif sums is not None:
        raise ValueError("sums without mask")
    rc = _timed("conv_dgrad", d, dy.element_size(), lambda: lib.msfwsi_img3x3_dgrad(
        C.byref(d), _p(dy), _p(cc), _p(k1), _p(k2), _p(k3), _p(dc_out), _p(wpk), _p(dx), _p(mc), _p(msc), _p(msh), _p(act_out),
        _p(sums), nsh, _stream()),
        extra_elems=dx.numel() * ((mask is not None) + (act_out is not None) + (bnbwd is not None) + (dc_out is not None)),
        halo=True, dtype=dy.dtype,
        symbol_override=_img3_symbol(d, dt, 2 if bnbwd is not None else 0, True))
    if rc == -2:
        return False
    _lib.check(rc, "img3x3_dgrad")
    return True


# ------------------------------------------------------------------------------------------------
# validation metrics (csrc/metrics.hip)
# ------------------------------------------------------------------------------------------------
def seg_stats(logits, pred, target, num_classes: int, pred_shift: int, target_shift: int, ignore_index):
    """tp, fp, fn, tn [N, C] int64 of a multiclass segmentation; `logits` [N, nch, ...] (argmax inside) or `pred` [N, ...]"""
    lib = _lib.load()
    N = target.shape[0]
    L = target.numel() // N
    _req(target, "target", torch.int64)
    nch, ldt = 0, 0
    if logits is not None:
        _req(logits, "logits")
        nch = logits.shape[1]
        ldt = dt_of(logits)
        if logits.shape[0] != N or logits.numel() != N * nch * L:
            raise ValueError(f"logits {tuple(logits.shape)} do not match target {tuple(target.shape)}")
    else:
        _req(pred, "pred", torch.int64, N * L)
    dev = target.device
    counts = torch.zeros(N, 4, num_classes, dtype=torch.int64, device=dev)
    outs = [torch.empty(N, num_classes, dtype=torch.int64, device=dev) for _ in range(4)]
    _lib.check(lib.msfwsi_seg_stats(ldt, _p(logits), nch, _p(pred), _p(target), N, L, int(num_classes), int(pred_shift),
                                    int(target_shift), int(ignore_index) if ignore_index is not None else 0,
                                    int(ignore_index is not None), _p(counts), *[_p(o) for o in outs], _stream()),
               "seg_stats")
    return tuple(outs)


def seg_scores(tp, fp, fn, tn, zero_division: float = 1.0):
    """fp64 [3 + 3C]: micro F1 / IoU / accuracy, then per-class F1, IoU, accuracy on the counts summed over images"""
    lib = _lib.load()
    N, Cn = tp.shape
    for nm, t in (("tp", tp), ("fp", fp), ("fn", fn), ("tn", tn)):
        _req(t, nm, torch.int64, N * Cn)
    out = torch.empty(3 + 3 * Cn, dtype=torch.float64, device=tp.device)
    _lib.check(lib.msfwsi_seg_scores(_p(tp), _p(fp), _p(fn), _p(tn), N, Cn, float(zero_division), _p(out), _stream()),
               "seg_scores")
    return out


def seg_scores_imagewise(tp, fp, fn, tn, zero_division: float = 1.0):
    """fp64 [6]: F1 / IoU / accuracy with smp's 'micro-imagewise' reduction, then with 'macro-imagewise'"""
    lib = _lib.load()
    N, Cn = tp.shape
    for nm, t in (("tp", tp), ("fp", fp), ("fn", fn), ("tn", tn)):
        _req(t, nm, torch.int64, N * Cn)
    out = torch.empty(6, dtype=torch.float64, device=tp.device)
    _lib.check(lib.msfwsi_seg_scores_imagewise(_p(tp), _p(fp), _p(fn), _p(tn), N, Cn, float(zero_division), _p(out),
                                               _stream()), "seg_scores_imagewise")
    return out


# ------------------------------------------------------------------------------------------------
# tiling / normalising front end (csrc/tiler.hip)
# ------------------------------------------------------------------------------------------------
def tile_views(img, grid: int, perm, boxes, flips, mean, std, size: int = 224, max_pixel: float = 255.0):
    """img uint8 [B,H,W,3] -> fp32 [B, grid*grid, 3, size, size]; perm int64 [B,K] / flips uint8 [B,K] optional,
    boxes int32 [B,K,4] = x0, y0, w, h inside each block"""
    lib = _lib.load()
    B, H, W, ch = img.shape
    K = grid * grid
    _req(img, "img", torch.uint8, B * H * W * 3)
    if ch != 3 or H % grid or W % grid:
        raise ValueError(f"expected [B,H,W,3] with H, W divisible by {grid}, got {tuple(img.shape)}")
    _req(boxes, "boxes", torch.int32, B * K * 4)
    _opt(perm, "perm", torch.int64, B * K)
    _opt(flips, "flips", torch.uint8, B * K)
    # the boxes are validated where they are drawn, on the host (data.DeviceTiler.view): here it would be a read-back
    mean_t = torch.tensor(list(mean), dtype=torch.float32)
    std_t = torch.tensor(list(std), dtype=torch.float32)
    out = torch.empty(B, K, 3, size, size, dtype=torch.float32, device=img.device)
    import ctypes as C_

    m = (C_.c_float * 3)(*mean_t.tolist())
    s = (C_.c_float * 3)(*std_t.tolist())
    _lib.check(lib.msfwsi_tile_views(_p(img), B, H, W, int(grid), _p(perm), _p(boxes), _p(flips),
                                     C_.cast(m, C_.c_void_p), C_.cast(s, C_.c_void_p), float(max_pixel), int(size),
                                     _p(out), _stream()), "tile_views")
    return out


def tile_crops_u8(img, grid: int, perm, boxes, size: int = 224):
    """img uint8 [B,H,W,3] -> uint8 [B, grid*grid, size, size, 3]: crop + bilinear resize of tile_views alone"""
    lib = _lib.load()
    B, H, W, ch = img.shape
    K = grid * grid
    _req(img, "img", torch.uint8, B * H * W * 3)
    if ch != 3 or H % grid or W % grid:
        raise ValueError(f"expected [B,H,W,3] with H, W divisible by {grid}, got {tuple(img.shape)}")
    _req(boxes, "boxes", torch.int32, B * K * 4)
    _opt(perm, "perm", torch.int64, B * K)
    out = torch.empty(B, K, size, size, 3, dtype=torch.uint8, device=img.device)
    _lib.check(lib.msfwsi_tile_crops_u8(_p(img), B, H, W, int(grid), _p(perm), _p(boxes), int(size), _p(out), _stream()),
               "tile_crops_u8")
    return out


def gray_sum(img):
    """img uint8 [N,H,W,3] -> fp64 [N]: sum of the 8-bit gray value over each image"""
    lib = _lib.load()
    N, H, W, ch = img.shape
    _req(img, "img", torch.uint8, N * H * W * 3)
    sums = torch.zeros(N, dtype=torch.float64, device=img.device)
    _lib.check(lib.msfwsi_gray_sum(_p(img), N, H, W, _p(sums), _stream()), "gray_sum")
    return sums


def color_stage(img, op, factor, gray_sums=None, out=None):
    """one colour adjustment per image (op int32 [N], factor fp64 [N]); in place unless `out` is given"""
    lib = _lib.load()
    N, H, W, ch = img.shape
    _req(img, "img", torch.uint8, N * H * W * 3)
    _req(op, "op", torch.int32, N)
    _opt(factor, "factor", torch.float64, N)
    _opt(gray_sums, "gray_sums", torch.float64, N)
    out = img if out is None else out
    _req(out, "out", torch.uint8, N * H * W * 3)
    _lib.check(lib.msfwsi_color_stage(_p(img), _p(out), N, H, W, _p(op), _p(factor), _p(gray_sums), _stream()),
               "color_stage")
    return out


def blur_sharpen(img, kind, ksize, taps):
    """kind int32 [N] (0 copy, 1 Gaussian blur, 2 sharpen), ksize int32 [N], taps fp32 [N,32] -> new uint8 image"""
    lib = _lib.load()
    N, H, W, ch = img.shape
    _req(img, "img", torch.uint8, N * H * W * 3)
    _req(kind, "kind", torch.int32, N)
    _req(ksize, "ksize", torch.int32, N)
    _req(taps, "taps", torch.float32, N * 32)
    out = torch.empty_like(img)
    tmp = torch.empty(N, H, W, 3, dtype=torch.float32, device=img.device)
    _lib.check(lib.msfwsi_blur_sharpen(_p(img), _p(out), _p(tmp), N, H, W, _p(kind), _p(ksize), _p(taps), _stream()),
               "blur_sharpen")
    return out


def inverse_perm(perm):
    """argsort of each row of a permutation matrix int64 [rows, K]"""
    lib = _lib.load()
    rows, K = perm.shape
    _req(perm, "perm", torch.int64)
    inv = torch.empty_like(perm)
    _lib.check(lib.msfwsi_inverse_perm(_p(perm), _p(inv), rows, K, _stream()), "inverse_perm")
    return inv


# ------------------------------------------------------------------------------------------------
# U-Net decoder pieces / Dice loss (csrc/unet.hip)
# ------------------------------------------------------------------------------------------------
def upcat_fwd(x, skip, out):
    """out [N,2h,2w,Cx+Cs] = [nearest-x2(x) | skip]"""
    lib = _lib.load()
    N, h, w, Cx = x.shape
    Cs = skip.shape[-1] if skip is not None else 0
    _req(x, "x")
    _opt(skip, "skip", x.dtype, N * 4 * h * w * Cs)
    _req(out, "out", x.dtype, N * 4 * h * w * (Cx + Cs))
    _lib.check(lib.msfwsi_upcat_fwd(dt_of(x), _p(x), _p(skip), _p(out), N, h, w, Cx, Cs, _stream()), "upcat_fwd")
    return out


def upcat_bwd(dout, dx, dskip):
    lib = _lib.load()
    N, h, w, Cx = dx.shape
    Cs = dout.shape[-1] - Cx
    _req(dout, "dout", None, N * 4 * h * w * (Cx + Cs))
    _req(dx, "dx", dout.dtype)
    _opt(dskip, "dskip", dout.dtype, N * 4 * h * w * Cs)
    _lib.check(lib.msfwsi_upcat_bwd(dt_of(dout), _p(dout), _p(dx), _p(dskip), N, h, w, Cx, Cs, _stream()), "upcat_bwd")


def crop(x, out, y0: int, x0: int, backward: bool = False):
    """out = x[:, y0:y0+ch, x0:x0+cw, :]; backward: x[window] += out"""
    lib = _lib.load()
    N, H, W, Cn = x.shape
    _, ch, cw, _ = out.shape
    _req(x, "x")
    _req(out, "out", x.dtype, N * ch * cw * Cn)
    _lib.check(lib.msfwsi_crop(dt_of(x), _p(x), _p(out), N, H, W, Cn, int(y0), int(x0), ch, cw, int(bool(backward)),
                               _stream()), "crop")
    return out


def nhwc_to_nchw(x, C1: int):
    """fp32 NCHW copy of the first C1 channels of an NHWC storage tensor"""
    lib = _lib.load()
    N, H, W, CP = x.shape
    _req(x, "x")
    y = torch.empty(N, C1, H, W, dtype=torch.float32, device=x.device)
    _lib.check(lib.msfwsi_nhwc_to_nchw(dt_of(x), _p(x), _p(y), N, int(C1), H * W, CP, _stream()), "nhwc_to_nchw")
    return y


def dice_loss(logits, target, C1: int, classes, weight: float, loss_accum, dlogits=None, grad_scale=None,
              eps: float = 1e-7, smooth: float = 0.0):
    """multiclass soft Dice from NHWC logits [N,H,W,CP] (C1 real channels), target int64 [N,H,W]; loss_accum fp64[1] +="""
    lib = _lib.load()
    N, H, W, CP = logits.shape
    M = N * H * W
    _req(logits, "logits")
    _req(target, "target", torch.int64, M)
    _req(loss_accum, "loss_accum", torch.float64, 1)
    _opt(dlogits, "dlogits", logits.dtype, M * CP)
    _opt(grad_scale, "grad_scale", torch.float32, 1)
    mask = 0
    for c in classes:
        if not 0 <= int(c) < C1:
            raise ValueError(f"class {c} outside the {C1} logit channels")
        mask |= 1 << int(c)
    sums = zeros((3, C1), torch.float64, logits.device)
    coef = torch.empty(2, C1, dtype=torch.float32, device=logits.device)
    _lib.check(lib.msfwsi_dice_loss(dt_of(logits), _p(logits), _p(target), M, int(C1), CP, mask, float(eps),
                                    float(smooth), float(weight), _p(sums), _p(loss_accum), _p(coef), _p(grad_scale),
                                    _p(dlogits), _stream()), "dice_loss")
