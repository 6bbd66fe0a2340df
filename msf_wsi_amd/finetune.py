"""The fine-tune step (BASELINE config 5, row f2 of SURVEY.md 8f) as one fused schedule on MI355X.

`FinetuneStep.step(images, masks)` performs one iteration of the reference's fine-tune loop
(tools/ssl_finetune.py:422-462): forward of HookNet (src/models/hooknet.py:246-252: context U-Net, hook, target
U-Net), the Dice loss on both logit maps weighted (1 - lam, lam) (:433-436), the per-step confusion counts of the target
prediction (:440-447), backward, GradScaler protocol and Adam over ALL parameters (:289, :455-458; the encoders are not
frozen) -- every arithmetic operation in the hand-written gfx950 kernels, the logits never leaving their NHWC
storage layout between the segmentation head, the loss and the backward, no host synchronisation inside the step.
`validate` is the evaluation loop's arithmetic (:497-530): eval-mode forward in chunks of 128 tile pairs and the
confusion counts of the target prediction against the target mask.

The decoder / Dice / metric arithmetic lives in segmentation_models_pytorch (third party, outside the reference tree,
absent here): its published algorithm is restated by the test-side checker -- parity unpinned.
"""
from __future__ import annotations

import math
from typing import Optional, Sequence, Tuple

import torch
import torch.nn as nn

from . import _lib
from . import kernels as kn
from .dist import FlatGroups, GradReducer, world_size
from .engine import Engine
from .train import FlatAdamScaler, _FlatGradStore
from .unet_engine import UnetEngine


class FinetuneStep(FlatAdamScaler):
    def __init__(self, model: nn.Module, lr: float = 1e-3, batch_size: int = 64, lam: float = 1.0,
                 classes: Optional[Sequence[int]] = None, dtype: torch.dtype = torch.bfloat16,
                 use_scaler: Optional[bool] = None, init_scale: float = 65536.0, process_group=None,
                 sync_bn: bool = True):
        """lam: weight of the target branch's loss (reference default 1, ssl_finetune.py:690); classes: the class
        indices the Dice loss averages over (reference: 1..n, the background channel 0 excluded, :287-288)"""
        _lib.load()
        dev = next(model.parameters()).device
        if dev.type != "cuda":
            raise _lib.MsfwsiHipError("FinetuneStep needs the model on a HIP device (model.cuda()); no CPU path")
        if dtype not in (torch.float32, torch.bfloat16, torch.float16):
            raise _lib.MsfwsiHipError(f"unsupported compute dtype {dtype}")
        self.model, self.dtype, self.device, self.group = model, dtype, dev, process_group
        self.lam = float(lam)
        self.n_logits = model.context_branch.segmentation_head[0].out_channels
        self.classes = list(range(1, self.n_logits)) if classes is None else [int(c) for c in classes]
        self.init_lr = lr * math.sqrt(batch_size) / math.sqrt(64)  # ssl_finetune.py:178
        self.lrs, self.eps, self.betas = [self.init_lr], [1e-8], (0.9, 0.999)
        self.flats = FlatGroups(model, lowp_dtype=None if dtype == torch.float32 else dtype, prefixes=("",))
        self.engine = Engine(process_group=process_group, sync_bn=sync_bn)
        model._engine = self.engine
        self.ue = UnetEngine(self.engine)
        self._register_lowp_weights()
        self.grads = _FlatGradStore(self.flats)
        self.reducer = GradReducer(self.flats, process_group)
        self._init_optimizer_state(use_scaler, init_scale)
        self.epoch_meter = torch.zeros(2, dtype=torch.float64, device=dev)

    # ---------------------------------------------------------------------------------------
    def forward_loss(self, images, masks, want_grad: bool = True):
        """images = (context [B,3,H,W], target [B,3,H,W]); masks = (context [B,H,W], target [B,H,W]) integer labels.
        Returns (branch records, dLoss/dlogits per branch or None).  loss_accum holds the loss afterwards."""
        ue, model, dtype = self.ue, self.model, self.dtype
        crec, hook = ue.branch_forward(model.context_branch, images[0], dtype)
        trec, _ = ue.branch_forward(model.target_branch, images[1], dtype, hook_in=hook)
        self.loss_accum.zero_()
        ls = self.scale if self.use_scaler else None
        dls = []
        for rec, m, wgt in ((crec, masks[0], 1.0 - self.lam), (trec, masks[1], self.lam)):
            dl = torch.empty_like(rec.logits) if want_grad else None
            kn.dice_loss(rec.logits, m.long().contiguous(), self.n_logits, self.classes, wgt, self.loss_accum, dlogits=dl,
                         grad_scale=ls)
            dls.append(dl)
        return (crec, trec), dls

    def step(self, images, masks) -> Tuple[torch.Tensor, Tuple[torch.Tensor, ...]]:
        """one optimisation step; returns (device-resident fp64 loss, (tp, fp, fn, tn) of the target prediction
        [B, n_classes] int64 -- what the loop appends to tp_all ... tn_all, ssl_finetune.py:440-453)"""
        bs = images[0].shape[0]
        self.model.train()  # the reference's train() does so every epoch (ssl_finetune.py:419); validate() leaves eval mode
        self.engine.reset_counters()
        self.flats.zero_grads()
        kn.ARENA.begin_step(self.device)
        try:
            (crec, trec), (dlc, dlt) = self.forward_loss(images, masks, want_grad=True)
            loss = self.loss_accum.clone()
            # pred_mask = argmax(target logits) - 1 against masks[1] - 1, ignore_index = -1 (:440-447), from the logits
            stats = kn.seg_stats(kn.nhwc_to_nchw(trec.logits, self.n_logits), None, masks[1].long().contiguous(),
                                 self.n_logits - 1, -1, -1, -1)
            dctx = self.ue.branch_backward(self.model.target_branch, trec, dlt, self.grads, self.dtype, extra_head=128)
            self.ue.branch_backward(self.model.context_branch, crec, dlc, self.grads, self.dtype, dhook=dctx)
        finally:
            kn.ARENA.end_step()
        self.reducer.launch(0)
        self.reducer.wait()
        self.engine.close_counters()
        self.optimizer_step()
        self.epoch_meter[0] += loss[0] * bs
        self.epoch_meter[1] += bs
        return loss, stats

    def epoch_loss(self) -> float:
        m = self.epoch_meter.clone()
        if world_size(self.group) > 1:
            import torch.distributed as dist

            dist.all_reduce(m, group=self.group)
        self.epoch_meter.zero_()
        return float(m[0] / m[1])

    @torch.no_grad()
    def validate(self, context_imgs: torch.Tensor, target_imgs: torch.Tensor, target_masks: torch.Tensor,
                 chunk: int = 128):
        """the evaluation loop's arithmetic for one slide (ssl_finetune.py:497-530): eval-mode forward of the tile pairs
        in chunks of `chunk`, argmax of the target logits and the confusion counts against the target masks.
        Returns (tp, fp, fn, tn) int64 [tiles, n_classes]; the model is left in eval mode like the reference's."""
        self.model.eval()
        outs = []
        for i in range(0, context_imgs.shape[0], chunk):
            crec, hook = self.ue.branch_forward(self.model.context_branch, context_imgs[i:i + chunk], self.dtype)
            trec, _ = self.ue.branch_forward(self.model.target_branch, target_imgs[i:i + chunk], self.dtype, hook_in=hook)
            outs.append(kn.seg_stats(kn.nhwc_to_nchw(trec.logits, self.n_logits), None,
                                     target_masks[i:i + chunk].long().contiguous(), self.n_logits - 1, -1, -1, -1))
        return tuple(torch.cat([o[k] for o in outs], 0) for k in range(4))

    def checkpoint(self, epoch: int) -> dict:
        sd = {"module." + k: v.detach().clone() for k, v in self.model.state_dict().items()}
        return {"epoch": epoch + 1, "arch": getattr(self.model, "encoder_name", "resnet18"), "state_dict": sd,
                "optimizer": self.optimizer_state_dict(), "scaler": self.scaler_state_dict()}
