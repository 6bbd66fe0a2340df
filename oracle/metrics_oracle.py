"""ORACLE -- test infrastructure only (nothing under msf_wsi_amd/ may import it).

numpy restatement of the segmentation metrics the reference's fine-tune / evaluation loops call
(tools/ssl_finetune.py:526-551, tools/evaluate.py:285-305):  smp.metrics.get_stats(mode="multiclass", ignore_index,
num_classes) and f1_score / iou_score / accuracy with reduction "micro", None and (ssl_finetune.py:319)
"micro-imagewise" / "macro-imagewise".

PARITY UNPINNED: the arithmetic lives in the third-party package `segmentation-models-pytorch>=0.3.2`
(/root/reference/environment.yml:26; version not pinned exactly, source not under /root/reference, package absent from
this image), and the reference has no test or fixture for it.  What is restated is the package's published algorithm
(functional._get_stats_multiclass / _fbeta_score / _iou_score / _accuracy, zero_division=1.0); the anchors are the
reference's call sites above (argument values: pred - 1, target - 1, ignore_index=-1)."""
import numpy as np


def get_stats_multiclass(output, target, num_classes, ignore_index=None):
    """output / target: integer arrays [N, ...] -> tp, fp, fn, tn int64 [N, num_classes]"""
    output = np.asarray(output).astype(np.int64)
    target = np.asarray(target).astype(np.int64)
    n = output.shape[0]
    output, target = output.reshape(n, -1).copy(), target.reshape(n, -1).copy()
    num_elements = output.shape[1]
    ignore_per_sample = np.zeros(n, dtype=np.int64)
    if ignore_index is not None:
        ignore = target == ignore_index
        output[ignore] = -1
        target[ignore] = -1
        ignore_per_sample = ignore.sum(1)
    tp = np.zeros((n, num_classes), dtype=np.int64)
    fp, fn, tn = tp.copy(), tp.copy(), tp.copy()

    def histc(v):  # torch.histc(bins=C, min=0, max=C-1) on integer-valued data: one bin per class, out-of-range dropped
        v = v[(v >= 0) & (v <= num_classes - 1)]
        return np.bincount(v, minlength=num_classes).astype(np.int64)

    for i in range(n):
        matched = np.where(output[i] == target[i], target[i], -1)
        t = histc(matched)
        f_p = histc(output[i]) - t
        f_n = histc(target[i]) - t
        tp[i], fp[i], fn[i] = t, f_p, f_n
        tn[i] = num_elements - t - f_p - f_n - ignore_per_sample[i]
    return tp, fp, fn, tn


def _div(num, den, zero_division=1.0):
    num, den = np.asarray(num, dtype=np.float64), np.asarray(den, dtype=np.float64)
    with np.errstate(divide="ignore", invalid="ignore"):
        out = num / den
    return np.where(den == 0, zero_division, out)


def scores(tp, fp, fn, tn, reduction, zero_division=1.0):
    """(f1, iou, accuracy); reduction "micro": over everything; None: elementwise on the given counts;
    "micro-imagewise": counts summed over classes per image, score per image, mean over images;
    "macro-imagewise": score per (image, class), mean over classes and images  (smp functional._compute_metric)"""
    if reduction == "micro":
        tp, fp, fn, tn = (np.asarray(x).sum() for x in (tp, fp, fn, tn))
    elif reduction == "micro-imagewise":
        tp, fp, fn, tn = (np.asarray(x).sum(1) for x in (tp, fp, fn, tn))
    tp, fp, fn, tn = (np.asarray(x, dtype=np.float64) for x in (tp, fp, fn, tn))
    res = (_div(2 * tp, 2 * tp + fn + fp, zero_division), _div(tp, tp + fp + fn, zero_division),
           _div(tp + tn, tp + fp + fn + tn, zero_division))
    if reduction in ("micro-imagewise", "macro-imagewise"):
        res = tuple(r.mean() for r in res)
    return res
