"""Execution of the fine-tune model (HookNet: two ResNet U-Nets coupled by a centre-crop hook) on MI355X -- row f2 of
SURVEY.md 8f, BASELINE config 5.  Schedules the hand-written HIP kernels for the forward and the hand-derived backward of

  * the encoders WITH their skip connections: smp's ResNetEncoder returns [x, relu(bn1(conv1)), layer1..4]
    (the reference calls it through smp.Unet, src/models/hooknet.py:143-145,196-198)
  * the U-Net decoders: per block nearest-x2 upsample + skip concat (one kernel), conv3x3-BN-ReLU twice
    (smp DecoderBlock; ContextUnetDecoder / TargetUnetDecoder.forward, hooknet.py:15-35,84-100)
  * the hook: context decoder block-1 output cropped [12:20, 12:20] and concatenated to the target encoder's head
    (hooknet.py:29-32, 92)
  * the segmentation heads (conv3x3 with bias to `classes` logits)

and exposes them to torch autograd as ONE node (`hooknet_apply`), so the reference's loop statements
(tools/ssl_finetune.py:441-458: criterion on both logit maps, scaler.scale(loss).backward(), scaler.step) run unchanged.
Encoder blocks, BatchNorm (train statistics / SyncBatchNorm exchange / eval mode), conv kernels and the weight-gradient
path are the pre-train engine's (msf_wsi_amd.engine); only the decoder-specific data movement is new (csrc/unet.hip).
There is no CPU / eager-torch fallback."""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Optional, Tuple

import torch
import torch.nn as nn

from . import _lib
from . import kernels as kn
from .engine import Engine, EncPass, GradStore, Unit, WeightStore, default_engine

HOOK_BLOCK, HOOK_LO, HOOK_HI = 1, 16 - 4, 16 + 4  # hooknet.py:29-32 ("hardcoded for hooknet")


@dataclass
class DecBlockRec:
    x_shape: Tuple[int, int, int, int]
    skip_c: int
    u1: Unit
    u2: Unit


@dataclass
class HeadRec:
    conv: nn.Conv2d
    x: torch.Tensor
    desc: object
    wpad: torch.Tensor
    Kp: int


@dataclass
class BranchRec:
    enc: EncPass
    a0: torch.Tensor
    blocks: List[DecBlockRec]
    head: HeadRec
    logits: torch.Tensor  # NHWC, channel-padded


def _ceil8(n: int) -> int:
    return (n + 7) // 8 * 8


class UnetEngine:
    def __init__(self, engine: Optional[Engine] = None):
        self.eng = engine or default_engine()

    # ---- encoder with skip maps -------------------------------------------------------------------------------------
    def encoder_maps(self, enc: nn.Module, x: torch.Tensor, dtype) -> Tuple[EncPass, List[torch.Tensor], torch.Tensor]:
        """(pass record, [layer1..4 output maps], stem activation) -- smp ResNetEncoder.forward's feature list without
        the input itself (which the decoders drop, hooknet.py:16,86)"""
        ps = self.eng.encoder_forward(enc, x, dtype, save=True)
        a0 = torch.empty_like(ps.stem.c)
        kn.bn_act(ps.stem.c, ps.stem.st.scale, ps.stem.st.shift, a0, relu=True)
        maps = [b.y_out for b in ps.blocks if b.stage_end]
        return ps, maps, a0

    # ---- decoder ----------------------------------------------------------------------------------------------------
    def block_forward(self, blk: nn.Module, x: torch.Tensor, skip: Optional[torch.Tensor], dtype):
        N, h, w, Cx = x.shape
        Cs = skip.shape[-1] if skip is not None else 0
        if skip is not None and tuple(skip.shape[:3]) != (N, 2 * h, 2 * w):
            raise RuntimeError(f"Sizes of tensors must match except in dimension 1. Expected size {2 * h} but got size "
                               f"{skip.shape[1]} (decoder skip connection)")
        cat = torch.empty(N, 2 * h, 2 * w, Cx + Cs, dtype=dtype, device=x.device)
        kn.upcat_fwd(x, skip, cat)
        u1 = self.eng._unit_fwd(blk.conv1[0], blk.conv1[1], True, cat, None, (N, 2 * h, 2 * w, Cx + Cs), dtype)
        u2 = self.eng._unit_fwd(blk.conv2[0], blk.conv2[1], True, u1.c, u1.st, (N, 2 * h, 2 * w, u1.c.shape[-1]), dtype)
        out = torch.empty_like(u2.c)
        kn.bn_act(u2.c, u2.st.scale, u2.st.shift, out, relu=True)
        return out, DecBlockRec((N, h, w, Cx), Cs, u1, u2)

    def block_backward(self, rec: DecBlockRec, d_out: torch.Tensor, grads: GradStore, dtype):
        """d_out: gradient of the block output relu(bn2(conv2(.))) (engine-owned, overwritten).  Returns (dx, dskip)."""
        eng, u1, u2 = self.eng, rec.u1, rec.u2
        dev = d_out.device
        C2 = u2.c.shape[-1]
        s2 = kn.new_stats(C2, 2, dev)
        g = d_out.view(-1, C2)
        kn.act_bwd_reduce(g, u2.c.view(-1, C2), u2.st.scale, u2.st.shift, g, s2)
        k = eng._bn_bwd_coeffs(s2, 2, 1, u2.bn, u2.st, grads)
        kn.bn_bwd_apply(g, u2.c.view(-1, C2), k[0], k[1], k[2], g)
        dc2 = d_out
        eng._unit_wgrad(u2, dc2, grads, dtype)
        C1 = u1.c.shape[-1]
        s1 = kn.new_stats(C1, 2, dev)
        da1 = eng._unit_dgrad(u2, dc2, dtype, mask=(u1.c, u1.st.scale, u1.st.shift), sums=s1)
        k1 = eng._bn_bwd_coeffs(s1, 2, 1, u1.bn, u1.st, grads)
        kn.bn_bwd_apply(da1, u1.c, k1[0], k1[1], k1[2], da1)
        eng._unit_wgrad(u1, da1, grads, dtype)
        dcat = eng._unit_dgrad(u1, da1, dtype)
        N, h, w, Cx = rec.x_shape
        dx = torch.empty(N, h, w, Cx, dtype=dtype, device=dev)
        dskip = torch.empty(N, 2 * h, 2 * w, rec.skip_c, dtype=dtype, device=dev) if rec.skip_c else None
        kn.upcat_bwd(dcat, dx, dskip)
        return dx, dskip

    # ---- segmentation head (output channels padded to whole 16-byte chunks) -------------------------------------------
    def head_forward(self, conv: nn.Conv2d, x: torch.Tensor, dtype) -> HeadRec:
        N, H, W, Cin = x.shape
        K = conv.out_channels
        Kp = _ceil8(K)
        R, S = conv.kernel_size
        phys = WeightStore.physical(conv.weight)  # fp32 [K][R][S][Cin]
        wpad = torch.empty(Kp, R, S, Cin, dtype=dtype, device=x.device)
        kn.pad_cast(phys, wpad, 1, phys.numel(), wpad.numel())  # rows K..Kp-1 are zero: their logits are never read
        bpad = None
        if conv.bias is not None:
            bpad = torch.empty(Kp, dtype=torch.float32, device=x.device)
            kn.pad_cast(conv.bias.data, bpad, 1, K, Kp)
        d = kn.conv_desc(dtype, N, H, W, Cin, Kp, R, S, conv.stride[0], conv.padding[0])
        logits = torch.empty(N, d.P, d.Q, Kp, dtype=dtype, device=x.device)
        kn.conv_fwd(d, x, wpad, logits, bias=bpad)
        return HeadRec(conv, x, d, wpad, Kp), logits

    def head_backward(self, rec: HeadRec, dlogits: torch.Tensor, grads: GradStore, dtype) -> torch.Tensor:
        conv, d = rec.conv, rec.desc
        K = conv.out_channels
        dw = kn.zeros((rec.Kp, d.R, d.S, d.C), torch.float32, dlogits.device)
        kn.conv_wgrad(d, rec.x, dlogits, dw)
        n = K * d.R * d.S * d.C
        kn.copy2d(dw, 0, n, grads.get(conv.weight), 0, n, 1, n, accumulate=True)
        if conv.bias is not None:
            cs = kn.zeros((rec.Kp,), torch.float64, dlogits.device)
            kn.colsum(dlogits, cs)
            kn.add_f64_to_f32(cs[:K], grads.get(conv.bias), 1.0)
        dx = torch.empty_like(rec.x)
        kn.conv_dgrad(d, dlogits, rec.wpad, dx)
        return dx

    # ---- one branch -------------------------------------------------------------------------------------------------
    def branch_forward(self, unet: nn.Module, x: torch.Tensor, dtype, hook_in: Optional[torch.Tensor] = None):
        """returns (BranchRec, hooked features or None)"""
        ps, maps, a0 = self.encoder_maps(unet.encoder, x, dtype)
        head = maps[3]
        if hook_in is not None:  # torch.cat([head, context_feats], dim=1), hooknet.py:92
            N, h, w, Ch = head.shape
            if tuple(hook_in.shape[:3]) != (N, h, w):
                raise RuntimeError(f"Sizes of tensors must match except in dimension 1. Expected size {h} but got size "
                                   f"{hook_in.shape[1]} (the hook crop {HOOK_LO}:{HOOK_HI} lines up with 256x256 inputs)")
            Cc = hook_in.shape[-1]
            cat = torch.empty(N, h, w, Ch + Cc, dtype=dtype, device=x.device)
            kn.copy2d(head, 0, Ch, cat, 0, Ch + Cc, N * h * w, Ch)
            kn.copy2d(hook_in, 0, Cc, cat, Ch, Ch + Cc, N * h * w, Cc)
            head = cat
        skips = [maps[2], maps[1], maps[0], a0]
        cur, recs, hook = head, [], None
        for i, blk in enumerate(unet.decoder.blocks):
            cur, r = self.block_forward(blk, cur, skips[i] if i < len(skips) else None, dtype)
            recs.append(r)
            if hook_in is None and i == HOOK_BLOCK and isinstance(unet, _context_type()):
                N, H, W, Cn = cur.shape
                if H < HOOK_HI or W < HOOK_HI:
                    raise RuntimeError(f"the context hook crops [{HOOK_LO}:{HOOK_HI}] of a {H}x{W} map "
                                       "(hooknet.py:29-32 is hard-coded for 256x256 inputs)")
                hook = torch.empty(N, HOOK_HI - HOOK_LO, HOOK_HI - HOOK_LO, Cn, dtype=dtype, device=x.device)
                kn.crop(cur, hook, HOOK_LO, HOOK_LO)
        hrec, logits = self.head_forward(unet.segmentation_head[0], cur, dtype)
        return BranchRec(ps, a0, recs, hrec, logits), hook

    def branch_backward(self, unet: nn.Module, rec: BranchRec, dlogits: torch.Tensor, grads: GradStore, dtype,
                        dhook: Optional[torch.Tensor] = None, extra_head: int = 0) -> Optional[torch.Tensor]:
        """dlogits NHWC padded (engine-owned).  dhook: gradient of the hooked crop (context branch).  Returns the
        gradient of the hooked features the target branch consumed (target branch) or None."""
        cur = self.head_backward(rec.head, dlogits, grads, dtype)
        dskips: List[Optional[torch.Tensor]] = [None] * 5
        for i in range(len(rec.blocks) - 1, -1, -1):
            if dhook is not None and i == HOOK_BLOCK:
                kn.crop(cur, dhook, HOOK_LO, HOOK_LO, backward=True)  # adjoint of the crop: += into its window
            cur, dskips[i] = self.block_backward(rec.blocks[i], cur, grads, dtype)
        dhead, dctx = cur, None
        if extra_head:
            N, h, w, Ct = cur.shape
            Ch = Ct - extra_head
            dhead = torch.empty(N, h, w, Ch, dtype=dtype, device=cur.device)
            dctx = torch.empty(N, h, w, extra_head, dtype=dtype, device=cur.device)
            kn.copy2d(cur, 0, Ct, dhead, 0, Ch, N * h * w, Ch)
            kn.copy2d(cur, Ch, Ct, dctx, 0, extra_head, N * h * w, extra_head)
        dmaps = [dskips[2], dskips[1], dskips[0], dhead]  # layer1..layer4 outputs
        self.eng.encoder_backward(rec.enc, [None] * 4, grads, dtype, dmaps=dmaps, dstem=dskips[3])
        return dctx


def _context_type():
    from .models.hooknet import ContextUnet

    return ContextUnet


# ----------------------------------------------------------------------------------------------------------------------
# autograd bridge
# ----------------------------------------------------------------------------------------------------------------------
class _HookNetFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, model, ue, dtype, x1, x2, *params):
        crec, hook = ue.branch_forward(model.context_branch, x1, dtype)
        trec, _ = ue.branch_forward(model.target_branch, x2, dtype, hook_in=hook)
        ctx.model, ctx.ue, ctx.dtype, ctx.recs, ctx.params = model, ue, dtype, (crec, trec), params
        K = model.context_branch.segmentation_head[0].out_channels
        return kn.nhwc_to_nchw(crec.logits, K), kn.nhwc_to_nchw(trec.logits, K)

    @staticmethod
    def backward(ctx, g_ctx, g_tgt):
        if ctx.recs is None:
            raise RuntimeError("the HookNet HIP node supports a single backward pass")
        (crec, trec), ctx.recs = ctx.recs, None
        model, ue, dtype = ctx.model, ctx.ue, ctx.dtype

        def to_engine(g, like):
            out = torch.empty_like(like)
            if g is None:
                out.zero_()
                return out
            kn.nchw_to_nhwc(g.detach().float().contiguous(), out, like.shape[-1])
            return out

        grads = GradStore()
        dctx = ue.branch_backward(model.target_branch, trec, to_engine(g_tgt, trec.logits), grads, dtype, extra_head=128)
        ue.branch_backward(model.context_branch, crec, to_engine(g_ctx, crec.logits), grads, dtype, dhook=dctx)
        return (None,) * 5 + tuple(grads.logical(p) for p in ctx.params)


def hooknet_apply(model: nn.Module, x1: torch.Tensor, x2: torch.Tensor):
    if not (x1.is_cuda and x2.is_cuda):
        raise _lib.MsfwsiHipError("HookNet runs only on a HIP device (no CPU path)")
    eng = getattr(model, "_engine", None) or default_engine()
    ue = UnetEngine(eng)
    dtype = eng.compute_dtype()
    params = list(model.parameters())
    K = model.context_branch.segmentation_head[0].out_channels
    if torch.is_grad_enabled() and any(p.requires_grad for p in params):
        return _HookNetFn.apply(model, ue, dtype, x1, x2, *params)
    crec, hook = ue.branch_forward(model.context_branch, x1, dtype)
    trec, _ = ue.branch_forward(model.target_branch, x2, dtype, hook_in=hook)
    return kn.nhwc_to_nchw(crec.logits, K), kn.nhwc_to_nchw(trec.logits, K)
