"""Drop-in for the reference's ``src/models/backbone.py``: the MSF-WSI multi-resolution SimSiam model.

Same constructor / forward signature, attribute names, module tree and state-dict keys as the reference
(src/models/backbone.py:34-222); the forward (two encoders x two views, jigsaw un-shuffle, 24 projector /
predictor MLPs, fuser concat) and its backward execute in hand-written gfx950 kernels via
:mod:`msf_wsi_amd.engine`.  ``forward`` returns autograd-connected tensors, so the reference training loop
(``tools/ssl_train.py:441-474``: torch cosine loss, ``scaler.scale(loss).backward()``, ``scaler.step``) runs
unchanged on top of it, as does DDP / SyncBatchNorm conversion.

Generalisation beyond the reference: ``inter_dim`` is ``[64,128,256,512] * block.expansion`` instead of the
hard-coded BasicBlock widths (backbone.py:67), so Bottleneck encoders (resnet50…) work; for resnet18/34
this is identical to the reference.
"""
from __future__ import annotations

import torch
import torch.nn as nn


def make_projector(in_dim, out_dim):
    """3-layer projector MLP (reference backbone.py:12-22): [Linear-BN-ReLU] x2, Linear-BN(affine=False)."""
    in_dim, out_dim = int(in_dim), int(out_dim)
    layers = []
    for _ in range(2):
        layers += [nn.Linear(in_dim, in_dim, bias=False), nn.BatchNorm1d(in_dim), nn.ReLU(inplace=True)]
    layers += [nn.Linear(in_dim, out_dim, bias=False), nn.BatchNorm1d(out_dim, affine=False)]
    return nn.Sequential(*layers)


def make_predictor(in_dim, out_dim):
    """2-layer predictor MLP (reference backbone.py:25-31): Linear-BN-ReLU, Linear(bias)."""
    in_dim, out_dim = int(in_dim), int(out_dim)
    return nn.Sequential(nn.Linear(in_dim, out_dim, bias=False), nn.BatchNorm1d(out_dim), nn.ReLU(inplace=True),
                         nn.Linear(out_dim, in_dim))


class MSFWSI(nn.Module):
    """MSF-WSI backbone; ``dim`` and ``pred_dim`` are accepted and unused, exactly as in the reference."""

    def __init__(self, base_encoder, scale, dim=2048, pred_dim=512, mask_ratio=0.5, use_checkpoint=False):
        super().__init__()
        self.K = int(scale ** 2)
        self.n_keep = int(self.K * (1 - mask_ratio))

        self.context_encoder = base_encoder(zero_init_residual=True, pretrained=True, return_features=True)
        self.target_encoder = base_encoder(zero_init_residual=True, pretrained=True, return_features=True)
        expansion = self.context_encoder.fc.in_features // 512
        self.context_encoder.fc = nn.Identity()
        self.target_encoder.fc = nn.Identity()

        self.inter_dim = torch.as_tensor([64, 128, 256, 512]) * expansion
        self.ms_inter_dim = self.inter_dim * (self.n_keep + 1)

        for prefix, dims in (("context", self.inter_dim), ("target", self.inter_dim), ("inter", self.ms_inter_dim)):
            setattr(self, f"{prefix}_projector", nn.ModuleList([make_projector(d, d) for d in dims]))
        for prefix, dims in (("context", self.inter_dim), ("target", self.inter_dim), ("inter", self.ms_inter_dim)):
            setattr(self, f"{prefix}_predictor",
                    nn.ModuleList([make_predictor(d, torch.div(d, 4, rounding_mode="floor")) for d in dims]))

        # The reference's --use-ac wraps every Conv2d/Linear in torch activation checkpointing
        # (backbone.py:106-127).  Here the engine reads this flag (Engine.model_forward): with it set, both target
        # passes run features-only in the forward and are re-run right before their backward ("targets" recompute
        # mode: 16/17 of the activation memory against one extra forward of the target stream); results are
        # identical.  The reference's side effect of re-initialising both stem convs is kept.
        self.use_checkpoint = bool(use_checkpoint)
        if use_checkpoint:
            for enc in (self.context_encoder, self.target_encoder):
                enc.conv1 = nn.Conv2d(3, 64, kernel_size=7, stride=2, padding=3, bias=False)
                enc.conv1.weight.data = enc.conv1.weight.data.contiguous(memory_format=torch.channels_last)

    def forward(self, x1, x2, jigsaw_idx=None):
        """x1, x2: (context_images [B,3,H,W], target_images [B*K,3,H,W]) of the two views;
        jigsaw_idx: two int64 [B,K] inverse permutations.  Returns the reference's 3 x 4 x 4 nested tuple
        ((ctx_p1, ctx_p2, ctx_z1, ctx_z2), (tgt ...), (fuser ...)), every z detached."""
        from .. import engine

        return engine.msfwsi_apply(self, x1, x2, jigsaw_idx)
