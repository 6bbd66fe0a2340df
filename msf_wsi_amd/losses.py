"""Losses of the fine-tune loop (row f2 of SURVEY.md 8f) on MI355X, behind the names the reference uses:

    criterion = smp.losses.DiceLoss(smp.losses.MULTICLASS_MODE, classes=cls_idx, from_logits=True)
                                                                                    tools/ssl_finetune.py:287-288
    loss = (1 - lam) * criterion(context_logits_mask, masks[0]) + lam * criterion(target_logits_mask, masks[1])   :444-447

`segmentation_models_pytorch` is a third-party dependency outside the reference tree (absent from this image): its
published algorithm (losses/dice.py + losses/_functional.soft_dice_score) is restated -- parity unpinned.  The reduction,
the loss value and d loss / d logits are computed by the HIP kernels of csrc/unet.hip; there is no CPU path."""
from __future__ import annotations

from typing import List, Optional

import torch

from . import _lib
from . import kernels as kn

BINARY_MODE, MULTICLASS_MODE, MULTILABEL_MODE = "binary", "multiclass", "multilabel"


class _DiceFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, target, classes, eps, smooth):
        N, C1, H, W = logits.shape
        CP = (C1 + 7) // 8 * 8
        nhwc = torch.empty(N, H, W, CP, dtype=torch.float32, device=logits.device)
        kn.nchw_to_nhwc(logits.detach().float().contiguous(), nhwc, CP)
        loss = torch.zeros(1, dtype=torch.float64, device=logits.device)
        kn.dice_loss(nhwc, target, C1, classes, 1.0, loss, eps=eps, smooth=smooth)
        ctx.save_for_backward(nhwc, target)
        ctx.meta = (C1, classes, eps, smooth, logits.dtype)
        return loss[0].float()

    @staticmethod
    def backward(ctx, g):
        nhwc, target = ctx.saved_tensors
        C1, classes, eps, smooth, dt = ctx.meta
        dl = torch.empty_like(nhwc)
        scratch = torch.zeros(1, dtype=torch.float64, device=nhwc.device)
        kn.dice_loss(nhwc, target, C1, classes, 1.0, scratch, dlogits=dl, grad_scale=g.detach().float().reshape(1).contiguous(),
                     eps=eps, smooth=smooth)
        return kn.nhwc_to_nchw(dl, C1).to(dt), None, None, None, None


class DiceLoss(torch.nn.Module):
    """smp.losses.DiceLoss for mode="multiclass", from_logits=True (the reference's only use)"""

    def __init__(self, mode: str, classes: Optional[List[int]] = None, log_loss: bool = False, from_logits: bool = True,
                 smooth: float = 0.0, ignore_index: Optional[int] = None, eps: float = 1e-7):
        super().__init__()
        if mode != MULTICLASS_MODE or log_loss or not from_logits or ignore_index is not None:
            raise NotImplementedError("only DiceLoss(MULTICLASS_MODE, classes=..., from_logits=True) is used by MSF-WSI")
        self.mode, self.classes, self.smooth, self.eps = mode, classes, float(smooth), float(eps)

    def forward(self, y_pred: torch.Tensor, y_true: torch.Tensor) -> torch.Tensor:
        if not y_pred.is_cuda:
            raise _lib.MsfwsiHipError("DiceLoss runs only on a HIP device (no CPU path)")
        if y_true.size(0) != y_pred.size(0):
            raise AssertionError("y_true.size(0) == y_pred.size(0)")
        C1 = y_pred.shape[1]
        classes = list(range(C1)) if self.classes is None else list(self.classes)
        return _DiceFn.apply(y_pred, y_true.long().contiguous(), classes, self.eps, self.smooth)
