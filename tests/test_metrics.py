"""Row f4 (validation metrics): the numpy oracle on hand-checkable cases (CPU), the HIP kernels against the oracle
bit for bit on the integer counts (GPU).  Third-party arithmetic (smp.metrics): parity unpinned, see oracle header."""
import numpy as np
import pytest
import torch


def test_oracle_on_a_hand_checked_case():
    from oracle import metrics_oracle as mo

    # one image, 8 pixels, 3 classes; the reference passes pred - 1 / target - 1 with ignore_index = -1
    pred = np.array([[0, 1, 2, 3, 3, 1, 0, 2]]) - 1     # -1 = "background predicted"
    tgt = np.array([[0, 1, 2, 3, 1, 1, 2, 0]]) - 1      # -1 = unlabelled -> ignored
    tp, fp, fn, tn = mo.get_stats_multiclass(pred, tgt, 3, ignore_index=-1)
    # valid pixels (target >= 0): idx 1..6; matches: idx1 (0), idx2 (1), idx3 (2), idx5 (0) -> tp = [2,1,1]
    assert tp.tolist() == [[2, 1, 1]]
    # predictions among valid pixels: cls0 at idx1,5 ; cls1 at idx2 ; cls2 at idx3,4 ; idx6 predicted background
    assert fp.tolist() == [[0, 0, 1]]
    # targets: cls0 at idx1,4,5 ; cls1 at idx2,6 ; cls2 at idx3
    assert fn.tolist() == [[1, 1, 0]]
    assert tn.tolist() == [[8 - 2 - 0 - 1 - 2, 8 - 1 - 0 - 1 - 2, 8 - 1 - 1 - 0 - 2]]
    f1, iou, acc = mo.scores(tp, fp, fn, tn, "micro")
    assert abs(float(f1) - 8 / 11) < 1e-15 and abs(float(iou) - 4 / 7) < 1e-15 and abs(float(acc) - 15 / 18) < 1e-15
    f1c, iouc, accc = mo.scores(tp.sum(0), fp.sum(0), fn.sum(0), tn.sum(0), None)
    assert np.allclose(f1c, [4 / 5, 2 / 3, 2 / 3]) and np.allclose(iouc, [2 / 3, 1 / 2, 1 / 2])
    # image-wise reductions (the training epoch's reduction="micro-imagewise", ssl_finetune.py:319) on two images
    tp2 = np.array([[2, 1, 1], [0, 0, 3]]); fp2 = np.array([[0, 0, 1], [1, 0, 0]])
    fn2 = np.array([[1, 1, 0], [0, 0, 1]]); tn2 = np.array([[3, 4, 4], [3, 4, 0]])
    f1i, ioui, _ = mo.scores(tp2, fp2, fn2, tn2, "micro-imagewise")
    assert abs(float(f1i) - 0.5 * (8 / 11 + 6 / 8)) < 1e-15 and abs(float(ioui) - 0.5 * (4 / 7 + 3 / 5)) < 1e-15
    f1m, _, _ = mo.scores(tp2, fp2, fn2, tn2, "macro-imagewise")
    assert abs(float(f1m) - (4 / 5 + 2 / 3 + 2 / 3 + 0 + 1 + 6 / 7) / 6) < 1e-15
    # a class that never occurs: 0/0 -> zero_division = 1.0 (smp default)
    tp0, fp0, fn0, tn0 = mo.get_stats_multiclass(np.array([[0, 0]]), np.array([[0, 0]]), 2)
    assert float(mo.scores(tp0.sum(0), fp0.sum(0), fn0.sum(0), tn0.sum(0), None)[0][1]) == 1.0


@pytest.mark.gpu
@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16, torch.float16])
@pytest.mark.parametrize("shape", [(3, 5, 37, 41), (2, 4, 256, 256), (130, 2, 8, 8), (1, 21, 64, 64)])
def test_seg_stats_match_oracle(hip_lib, dt, shape):
    """argmax + confusion counts + scores; N images, C classes (+ background channel 0), ragged sizes, ignored pixels"""
    import msf_wsi_amd.metrics as metrics
    from oracle import metrics_oracle as mo

    N, C, H, W = shape
    g = torch.Generator().manual_seed(N * 1000 + C)
    logits = torch.randn(N, C + 1, H, W, generator=g).to(dt)
    logits[:, :, :2, :3] = 0.5  # exact ties: the first maximum wins (torch.argmax)
    target = torch.randint(0, C + 1, (N, H, W), generator=g)   # 0 = unlabelled -> ignored after "- 1"
    if N > 1:
        target[1] = 0                                          # a fully ignored image
    ref_pred = torch.argmax(logits.float(), dim=1)
    want = mo.get_stats_multiclass(ref_pred.numpy() - 1, target.numpy() - 1, C, ignore_index=-1)
    # 1) the reference's statements one by one
    tp, fp, fn, tn = metrics.get_stats(ref_pred.cuda() - 1, target.cuda() - 1, mode="multiclass", ignore_index=-1,
                                       num_classes=C)
    # 2) fused from the logits (the prediction map never exists)
    fused = metrics.get_stats_from_logits(logits.cuda(), target.cuda(), C)
    torch.cuda.synchronize()
    for got, got2, ref in zip((tp, fp, fn, tn), fused, want):
        assert got.dtype == torch.int64 and tuple(got.shape) == (N, C)
        assert np.array_equal(got.cpu().numpy(), ref) and np.array_equal(got2.cpu().numpy(), ref)
    for fn_, idx in ((metrics.f1_score, 0), (metrics.iou_score, 1), (metrics.accuracy, 2)):
        micro = fn_(tp, fp, fn, tn, reduction="micro")
        per_class = fn_(tp.sum(0), fp.sum(0), fn.sum(0), tn.sum(0), reduction=None)
        torch.cuda.synchronize()
        assert abs(float(micro) - float(mo.scores(*want, "micro")[idx])) < 1e-14
        ref_c = mo.scores(*[w.sum(0) for w in want], None)[idx]
        assert np.allclose(per_class.cpu().numpy(), ref_c, rtol=0, atol=1e-14)
        for red in ("micro-imagewise", "macro-imagewise"):
            got_i = fn_(tp, fp, fn, tn, reduction=red)
            torch.cuda.synchronize()
            assert abs(float(got_i) - float(mo.scores(*want, red)[idx])) < 1e-13, red


@pytest.mark.gpu
def test_metrics_reject_cpu_tensors(hip_lib):
    import msf_wsi_amd.metrics as metrics
    from msf_wsi_amd._lib import MsfwsiHipError

    with pytest.raises(MsfwsiHipError):
        metrics.get_stats(torch.zeros(1, 4, dtype=torch.long), torch.zeros(1, 4, dtype=torch.long), mode="multiclass",
                          num_classes=2)
