"""Whole-model 16-bit gates on the DEFAULT configuration of the library.

tests/test_lowp_parity_gpu.py runs with `msfwsi_set_tuning(15, 1)` (conftest.reproducible_sums: one workgroup per
weight-gradient tile, every value repeats bit for bit) so that each statistical gate has one outcome per build.  bench.py
and every user run the default: pixel splits that add their partial weight gradients with fp32 atomics in arrival order,
on which the ResNet-50-derived bf16 gradients move by ~1e-2 between two runs of the same step.  These tests
hold THAT path to the same yardstick -- the reference under autocast -- three runs in a row, all three inside the
allowance: a default-path defect the reproducible mode hides (a race between splits, an atomics ordering that matters)
would show as a run outside it."""
import pytest
import torch

from helpers import build_case, load_golden
from test_lowp_parity_gpu import gate_lowp_step

pytestmark = pytest.mark.gpu  # (no reproducible_sums here: the default split-K / atomics path is the subject)


def _three_runs(case, dtype, what):
    _, man = load_golden(case)
    model = build_case(man).cuda().train()
    for run in range(3):
        gate_lowp_step(case, dtype, f"{what}, default path, run {run + 1}/3", model=model)


def test_default_path_lowp_r18(hip_lib):
    """ResNet-18 dual-stream, bf16 autocast, default split-K weight gradients: three runs, each within the
    reference-under-autocast allowance per output tensor, loss term and gradient tensor"""
    assert hip_lib.msfwsi_set_tuning(15, 0) == 0  # the default: no cap on the pixel splits
    _three_runs("r18_b16_s64_div", torch.bfloat16, "r18_b16_s64_div bf16")


def test_default_path_lowp_r50(hip_lib):
    """the ResNet-50-derived model bench.py runs (folded Bottleneck tails whose BatchNorm-backward coefficients inherit
    the atomics' jitter, 18432-wide fuser GEMMs), bf16, default path: three runs inside the allowance"""
    assert hip_lib.msfwsi_set_tuning(15, 0) == 0
    _three_runs("r50_b8_s64_div", torch.bfloat16, "r50_b8_s64_div bf16")
