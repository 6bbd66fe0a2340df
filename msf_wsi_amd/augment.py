"""Colour augmentations of the pre-train front end on the device (row f3 of SURVEY.md 8f).

The reference's DataLoader workers run, per sample and per view (tools/ssl_train.py:176-201, applied in
BcssPretrainDataset.__getitem__, src/utils/data/bcss.py:166-170):
    albu.ColorJitter(0.4, 0.4, 0.4, 0.1, p=0.8)
    albu.ToGray(p=0.2)
    albu.OneOf([albu.GaussianBlur(blur_limit=[19, 23], sigma_limit=[0.1, 2.0], p=0.5), albu.Sharpen(p=0.5)], p=0.5)
on the whole 1024x1024 tile for the target views and on the 224x224 crop for the context views.  `DeviceColorAug` draws
the same decisions on the host (a torch.Generator stands in for albumentations' use of Python's `random`) and applies
them to uint8 device images with the HIP kernels of csrc/augment.hip; `data.DeviceTiler.batch(..., color=...)` places it
where the reference's lists place it.  PARITY UNPINNED: albumentations / cv2 are absent from this image; their published
8-bit arithmetic is restated (see csrc/augment.hip for what is and is not reproduced).
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Optional

import torch

from . import kernels as kn

OP_NONE, OP_BRIGHTNESS, OP_CONTRAST, OP_SATURATION, OP_HUE, OP_GRAY = 0, 1, 2, 3, 4, 5
FILT_NONE, FILT_BLUR, FILT_SHARPEN = 0, 1, 2
MAX_TAPS = 32


@dataclass
class ColorDecisions:
    """per image, on the host: what albumentations' get_params / random draws would have decided"""
    jitter: torch.Tensor     # bool [N]           ColorJitter applies (p = 0.8)
    order: torch.Tensor      # int32 [N,4]        permutation of OP_BRIGHTNESS..OP_HUE
    factors: torch.Tensor    # float64 [N,5]      factors[n][op] (column 0 unused)
    gray: torch.Tensor       # bool [N]           ToGray applies (p = 0.2)
    filt: torch.Tensor       # int32 [N]          FILT_*
    ksize: torch.Tensor      # int32 [N]          Gaussian kernel size (odd, 19..23)
    sigma: torch.Tensor      # float64 [N]
    alpha: torch.Tensor      # float64 [N]        Sharpen
    lightness: torch.Tensor  # float64 [N]


def gaussian_taps(ksize: int, sigma: float) -> torch.Tensor:
    """cv2.getGaussianKernel(ksize, sigma): exp(-(i - c)^2 / (2 sigma^2)) normalised in float64, as fp32 taps"""
    x = torch.arange(ksize, dtype=torch.float64) - (ksize - 1) / 2.0
    k = torch.exp(-(x * x) / (2.0 * sigma * sigma))
    return (k / k.sum()).to(torch.float32)


def sharpen_matrix(alpha: float, lightness: float) -> torch.Tensor:
    """albumentations.Sharpen: (1 - alpha) * identity + alpha * [[-1,-1,-1],[-1,8+lightness,-1],[-1,-1,-1]]"""
    nochange = torch.tensor([[0, 0, 0], [0, 1, 0], [0, 0, 0]], dtype=torch.float64)
    effect = torch.tensor([[-1, -1, -1], [-1, 8 + lightness, -1], [-1, -1, -1]], dtype=torch.float64)
    return ((1 - alpha) * nochange + alpha * effect).to(torch.float32)


class DeviceColorAug:
    def __init__(self, brightness: float = 0.4, contrast: float = 0.4, saturation: float = 0.4, hue: float = 0.1,
                 p_jitter: float = 0.8, p_gray: float = 0.2, p_filter: float = 0.5, blur_limit=(19, 23),
                 sigma_limit=(0.1, 2.0), sharpen_alpha=(0.2, 0.5), sharpen_lightness=(0.5, 1.0), chunk: int = 32):
        self.ranges = {OP_BRIGHTNESS: (max(0.0, 1 - brightness), 1 + brightness),
                       OP_CONTRAST: (max(0.0, 1 - contrast), 1 + contrast),
                       OP_SATURATION: (max(0.0, 1 - saturation), 1 + saturation), OP_HUE: (-hue, hue)}
        self.p_jitter, self.p_gray, self.p_filter = p_jitter, p_gray, p_filter
        self.blur_limit, self.sigma_limit = tuple(blur_limit), tuple(sigma_limit)
        self.sharpen_alpha, self.sharpen_lightness = tuple(sharpen_alpha), tuple(sharpen_lightness)
        self.chunk = int(chunk)  # images per blur launch: bounds the fp32 scratch (chunk * H * W * 12 bytes)

    def decisions(self, N: int, gen: Optional[torch.Generator] = None) -> ColorDecisions:
        gen = gen or torch.Generator()
        u = lambda lo, hi: torch.empty(N, dtype=torch.float64).uniform_(lo, hi, generator=gen)
        factors = torch.zeros(N, 5, dtype=torch.float64)
        for op, (lo, hi) in self.ranges.items():
            factors[:, op] = u(lo, hi)
        order = torch.stack([torch.randperm(4, generator=gen) + 1 for _ in range(N)]).to(torch.int32)
        jitter = torch.rand(N, generator=gen) < self.p_jitter
        gray = torch.rand(N, generator=gen) < self.p_gray
        use = torch.rand(N, generator=gen) < self.p_filter
        blur = torch.rand(N, generator=gen) < 0.5  # OneOf: both members carry p = 0.5
        filt = torch.where(use, torch.where(blur, FILT_BLUR, FILT_SHARPEN), FILT_NONE).to(torch.int32)
        # GaussianBlur.get_params: randrange(lo, hi + 1); an even draw k becomes (k + 1) % (hi + 1)
        lo, hi = self.blur_limit
        k = torch.randint(lo, hi + 1, (N,), generator=gen)
        k = torch.where(k % 2 == 0, (k + 1) % (hi + 1), k).to(torch.int32)
        return ColorDecisions(jitter, order, factors, gray, filt, k, u(*self.sigma_limit), u(*self.sharpen_alpha),
                              u(*self.sharpen_lightness))

    def apply(self, img_u8: torch.Tensor, dec: ColorDecisions) -> torch.Tensor:
        """img_u8: uint8 [N,H,W,3] on the device -> a new uint8 tensor of the same shape"""
        if img_u8.dim() != 4 or img_u8.shape[-1] != 3 or img_u8.dtype != torch.uint8:
            raise ValueError(f"expected uint8 [N,H,W,3], got {img_u8.dtype} {tuple(img_u8.shape)}")
        N, H, W, _ = img_u8.shape
        if dec.jitter.numel() != N:
            raise ValueError("one decision per image")
        dev = img_u8.device
        img = img_u8.clone()
        # ---- ColorJitter: the four adjustments in each image's own order; images it skips carry OP_NONE
        for s in range(4):
            op = torch.where(dec.jitter, dec.order[:, s], OP_NONE).to(torch.int32)
            fac = dec.factors.gather(1, op.long().unsqueeze(1)).squeeze(1).contiguous()
            if not bool((op != OP_NONE).any()):
                continue
            sums = kn.gray_sum(img) if bool((op == OP_CONTRAST).any()) else None
            kn.color_stage(img, op.to(dev), fac.to(dev), sums)
        # ---- ToGray
        if bool(dec.gray.any()):
            kn.color_stage(img, torch.where(dec.gray, OP_GRAY, OP_NONE).to(torch.int32).to(dev), None, None)
        # ---- OneOf(GaussianBlur, Sharpen)
        if bool((dec.filt != FILT_NONE).any()):
            taps = torch.zeros(N, MAX_TAPS, dtype=torch.float32)
            for n in range(N):
                if int(dec.filt[n]) == FILT_BLUR:
                    ks = int(dec.ksize[n])
                    if ks > MAX_TAPS - 1 or ks % 2 == 0 or ks // 2 >= min(H, W):
                        raise ValueError(f"Gaussian kernel size {ks} does not fit ({MAX_TAPS - 1} taps, image {H}x{W})")
                    taps[n, :ks] = gaussian_taps(ks, float(dec.sigma[n]))
                elif int(dec.filt[n]) == FILT_SHARPEN:
                    taps[n, :9] = sharpen_matrix(float(dec.alpha[n]), float(dec.lightness[n])).reshape(-1)
            out = torch.empty_like(img)
            for i in range(0, N, self.chunk):
                j = min(N, i + self.chunk)
                out[i:j] = kn.blur_sharpen(img[i:j], dec.filt[i:j].to(dev), dec.ksize[i:j].to(dev), taps[i:j].to(dev))
            img = out
        return img
