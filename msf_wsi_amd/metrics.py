"""Validation metrics of the fine-tune / evaluation loops (row f4 of SURVEY.md 8f) on MI355X.

Mirrors the calls the reference makes (tools/ssl_finetune.py:526-551, tools/evaluate.py:285-305):

    pred_mask = torch.argmax(preds, dim=1)
    tp, fp, fn, tn = smp.metrics.get_stats(pred_mask.long() - 1, target_masks.long() - 1, mode="multiclass",
                                           ignore_index=-1, num_classes=len(class_names))
    smp.metrics.f1_score(tp, fp, fn, tn, reduction="micro")    iou_score(...)    accuracy(...)
    smp.metrics.f1_score(tp.sum(0), fp.sum(0), fn.sum(0), tn.sum(0), reduction=None)    ...
    smp.metrics.f1_score(tp, fp, fn, tn, reduction="micro-imagewise")          (training epoch, ssl_finetune.py:319)

with the same names and argument meaning, so `import msf_wsi_amd.metrics as metrics` stands in for `smp.metrics` in
those loops.  `segmentation_models_pytorch` is a third-party dependency that is not part of the reference tree (and is
absent from this image): its published algorithm is restated -- parity unpinned (the tests check it against a numpy
restatement of the same published algorithm).  Everything runs in the HIP kernels of csrc/metrics.hip on device
tensors; there is no CPU path."""
from __future__ import annotations

from typing import Optional, Tuple

import torch

from . import _lib
from . import kernels as kn


def get_stats(output: torch.Tensor, target: torch.Tensor, mode: str = "multiclass", ignore_index: Optional[int] = None,
              threshold=None, num_classes: Optional[int] = None) -> Tuple[torch.Tensor, ...]:
    """smp.metrics.get_stats for mode="multiclass": output / target are integer label maps [N, ...]; returns
    tp, fp, fn, tn as int64 [N, num_classes]"""
    if mode != "multiclass":
        raise NotImplementedError("only mode='multiclass' is used by MSF-WSI (ssl_finetune.py:530)")
    if threshold is not None:
        raise ValueError("threshold is a binary / multilabel argument")
    if num_classes is None:
        raise ValueError("num_classes is required for mode='multiclass'")
    if output.shape != target.shape:
        raise ValueError(f"output {tuple(output.shape)} and target {tuple(target.shape)} must have the same shape")
    if output.is_floating_point() or target.is_floating_point():
        raise ValueError("multiclass mode takes integer label maps")
    return kn.seg_stats(None, output.long(), target.long(), num_classes, 0, 0, ignore_index)


def get_stats_from_logits(logits: torch.Tensor, target: torch.Tensor, num_classes: int, ignore_index: Optional[int] = -1,
                          shift: int = -1) -> Tuple[torch.Tensor, ...]:
    """the reference's three statements in one pass over the logits [N, num_classes + 1, H, W]:
    get_stats(argmax(logits, 1) + shift, target + shift, ignore_index=..., num_classes=...) -- the int64 prediction map is
    never materialised (ssl_finetune.py:526-533)"""
    return kn.seg_stats(logits, None, target.long(), num_classes, shift, shift, ignore_index)


def _scores(tp, fp, fn, tn, zero_division: float):
    t2 = [t.reshape(-1, t.shape[-1]) if t.dim() > 1 else t.reshape(1, -1) for t in (tp, fp, fn, tn)]
    return kn.seg_scores(*[t.long().contiguous() for t in t2], zero_division)


def _pick(which: int, tp, fp, fn, tn, reduction, zero_division):
    s = _scores(tp, fp, fn, tn, zero_division)
    C = tp.shape[-1]
    if reduction == "micro":
        return s[which]
    if reduction is None or reduction == "none":
        if tp.dim() != 1:
            raise NotImplementedError("reduction=None is used on per-class totals (tp.sum(0)) by the reference")
        return s[3 + which * C:3 + (which + 1) * C]
    if reduction in ("micro-imagewise", "macro-imagewise"):
        if tp.dim() != 2:
            raise ValueError(f"reduction={reduction!r} needs per-image counts [N, C]")
        t4 = [t.long().contiguous() for t in (tp, fp, fn, tn)]
        return kn.seg_scores_imagewise(*t4, zero_division)[which + (3 if reduction == "macro-imagewise" else 0)]
    # smp also offers "macro", "weighted", "weighted-imagewise" (class_weights): no call site in the reference
    raise NotImplementedError(f"reduction={reduction!r} is not used by MSF-WSI (implemented: 'micro', None / 'none' on "
                              f"per-class totals, 'micro-imagewise', 'macro-imagewise')")


def f1_score(tp, fp, fn, tn, reduction: Optional[str] = None, class_weights=None, zero_division: float = 1.0):
    return _pick(0, tp, fp, fn, tn, reduction, zero_division)


def iou_score(tp, fp, fn, tn, reduction: Optional[str] = None, class_weights=None, zero_division: float = 1.0):
    return _pick(1, tp, fp, fn, tn, reduction, zero_division)


def accuracy(tp, fp, fn, tn, reduction: Optional[str] = None, class_weights=None, zero_division: float = 1.0):
    return _pick(2, tp, fp, fn, tn, reduction, zero_division)
