"""On-device batch front end of the pre-train step (row f3 of SURVEY.md 8f).

`BcssPretrainDataset.__getitem__` (reference src/utils/data/bcss.py:164-182; the PAIP / Camelyon datasets are identical)
produces, per 1024x1024 tile and per view, on DataLoader CPU workers:
    context view : RandomResizedCrop(224, scale 0.5..1) + flip + Normalize + ToTensor of the (colour-augmented) tile
    target view  : blockshaped 4x4 split into 256x256 blocks -> shuffled by jigsaw_idx = randperm(16) -> each block
                   through RandomResizedCrop(224) + flip + Normalize + ToTensor
    jigsaw_reverse_idx = argsort(jigsaw_idx)
up to 34 crops per sample.  `DeviceTiler.batch` does the same on the GPU from the uint8 tiles: the random decisions are
drawn on the host exactly where the reference draws them (torch.randperm for the jigsaw; albumentations'
RandomResizedCrop box law for the crops), the pixels never leave HBM.  The colour augmentations (ColorJitter, ToGray,
GaussianBlur / Sharpen, tools/ssl_train.py:176-201) are optional: `batch(..., color=DeviceColorAug())` runs them on
the device where the reference's lists place them -- on the whole tile BEFORE the split for the target views
(bcss.py:166-170), between the crop and the flip for the context views (msf_wsi_amd/augment.py; albumentations
arithmetic restated, unpinned); without it feed already colour-augmented tiles.

Output = exactly the batch contract the step consumes (tools/ssl_train.py:425-438):
    (ctx_v1, ctx_v2) fp32 [B,3,224,224], (tgt_v1, tgt_v2) fp32 [B*16,3,224,224] (flattened), [idx_v1, idx_v2] int64 [B,16]
"""
from __future__ import annotations

import math
from typing import Optional, Sequence, Tuple

import torch

from . import kernels as kn

MEAN, STD = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)


def random_resized_crop_box(h: int, w: int, gen: torch.Generator, scale=(0.5, 1.0), ratio=(3 / 4, 4 / 3)):
    """albumentations / torchvision RandomResizedCrop box law: 10 tries of area ~ U(scale) * h*w and log-uniform
    aspect ratio, else the centre crop at a clamped ratio; returns x0, y0, cw, ch"""
    area = h * w
    for _ in range(10):
        target = area * float(torch.empty(1).uniform_(scale[0], scale[1], generator=gen))
        logr = (math.log(ratio[0]), math.log(ratio[1]))
        ar = math.exp(float(torch.empty(1).uniform_(logr[0], logr[1], generator=gen)))
        cw, ch = int(round(math.sqrt(target * ar))), int(round(math.sqrt(target / ar)))
        if 0 < cw <= w and 0 < ch <= h:
            y0 = int(torch.randint(0, h - ch + 1, (1,), generator=gen))
            x0 = int(torch.randint(0, w - cw + 1, (1,), generator=gen))
            return x0, y0, cw, ch
    in_ratio = w / h
    if in_ratio < ratio[0]:
        cw, ch = w, int(round(w / ratio[0]))
    elif in_ratio > ratio[1]:
        ch, cw = h, int(round(h * ratio[1]))
    else:
        cw, ch = w, h
    return (w - cw) // 2, (h - ch) // 2, cw, ch


class DeviceTiler:
    def __init__(self, scale: int = 4, size: int = 224, mean: Sequence[float] = MEAN, std: Sequence[float] = STD,
                 crop_scale=(0.5, 1.0), flip_p: float = 0.5):
        self.grid, self.K, self.size = int(scale), int(scale) ** 2, int(size)
        self.mean, self.std, self.crop_scale, self.flip_p = tuple(mean), tuple(std), crop_scale, float(flip_p)

    def _decisions(self, B: int, K: int, bh: int, bw: int, gen: torch.Generator):
        boxes = torch.tensor([[random_resized_crop_box(bh, bw, gen, self.crop_scale) for _ in range(K)]
                              for _ in range(B)], dtype=torch.int32)
        flips = (torch.rand(B, K, generator=gen) < self.flip_p).to(torch.uint8)
        return boxes, flips

    def view(self, tiles_u8: torch.Tensor, grid: int, perm: Optional[torch.Tensor], boxes: torch.Tensor,
             flips: Optional[torch.Tensor]) -> torch.Tensor:
        """one augmented view of every tile: fp32 [B, grid*grid, 3, size, size]"""
        B, H, W, _ = tiles_u8.shape
        bh, bw = H // grid, W // grid
        b = boxes.view(-1, 4)
        if bool(((b[:, 0] < 0) | (b[:, 1] < 0) | (b[:, 2] <= 0) | (b[:, 3] <= 0) | (b[:, 0] + b[:, 2] > bw)
                 | (b[:, 1] + b[:, 3] > bh)).any()):
            raise ValueError("crop boxes must lie inside their block")
        dev = tiles_u8.device
        return kn.tile_views(tiles_u8, grid, perm.to(dev) if perm is not None else None, boxes.to(dev),
                             flips.to(dev) if flips is not None else None, self.mean, self.std, self.size)

    def batch(self, ctx_tiles_u8: Sequence[torch.Tensor], tgt_tiles_u8: Sequence[torch.Tensor],
              gen: Optional[torch.Generator] = None, color=None) -> Tuple[tuple, tuple, list]:
        """ctx_tiles_u8[v], tgt_tiles_u8[v]: uint8 [B,H,W,3] device tensors of view v -- raw tiles with
        color=augment.DeviceColorAug(), already colour-augmented ones without.
        Returns ((ctx_v1, ctx_v2), (tgt_v1, tgt_v2), [idx_v1, idx_v2]) as the step consumes them."""
        gen = gen or torch.Generator()
        ctx, tgt, idx = [], [], []
        for v in range(2):
            B, H, W, _ = tgt_tiles_u8[v].shape
            perm = torch.stack([torch.randperm(self.K, generator=gen) for _ in range(B)])  # bcss.py:171
            cb, cf = self._decisions(B, 1, H, W, gen)
            tb, tf = self._decisions(B, self.K, H // self.grid, W // self.grid, gen)
            ctx_in, tgt_in = ctx_tiles_u8[v], tgt_tiles_u8[v]
            if color is not None:
                # target_aug on the whole tile, before blockshaped (bcss.py:166-170)
                tgt_in = color.apply(tgt_in, color.decisions(B, gen))
                # context_aug: RandomResizedCrop, then the colour list, then flip + Normalize (ssl_train.py:176-196)
                crops = kn.tile_crops_u8(ctx_in, 1, None, cb.to(ctx_in.device), self.size).flatten(0, 1)
                ctx_in = color.apply(crops, color.decisions(B, gen))
                cb = torch.tensor([0, 0, self.size, self.size], dtype=torch.int32).repeat(B, 1, 1)  # exact copy
            ctx.append(self.view(ctx_in, 1, None, cb, cf).flatten(0, 1))
            tgt.append(self.view(tgt_in, self.grid, perm, tb, tf).flatten(0, 1))
            idx.append(kn.inverse_perm(perm.to(tgt_tiles_u8[v].device)))  # jigsaw_reverse_idx, bcss.py:172
        return tuple(ctx), tuple(tgt), idx
