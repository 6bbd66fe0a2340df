"""Drop-in for the reference's ``src/models/resnet.py`` (torchvision-style ResNet with multi-scale pooled
features), MI355X-native: the modules below only OWN parameters/buffers (same names, shapes, construction
and init order as the reference, so ``torch.manual_seed(s)`` reproduces its initialisation bit for bit and
state dicts are interchangeable); the arithmetic of ``forward`` runs in the hand-written HIP kernels of
``msf_wsi_amd/csrc`` through :mod:`msf_wsi_amd.engine`.

Reference interface mirrored here (file:line in the reference tree):
  * factories ``resnet18 … resnet152(pretrained=False, progress=True, **kwargs)``  src/models/resnet.py:278-330
  * ``ResNet(block, layers, num_classes, zero_init_residual, groups, width_per_group,
    replace_stride_with_dilation, norm_layer, return_features)``               src/models/resnet.py:143-205
  * ``forward`` with ``return_features=True`` returns ``(gap(x1), gap(x2), gap(x3), fc(gap(x4)))``
                                                                                src/models/resnet.py:232-256
Out of scope (never used by any reference script): grouped / wide variants, dilation, custom norm layers.
Conv weights are stored in torch ``channels_last`` memory format ([Cout][kh][kw][Cin] physically), which is
the operand layout of the gfx950 kernels; logical shapes and state-dict contents are unchanged.
"""
from __future__ import annotations

import os
from typing import Any, List, Optional, Sequence, Type

import torch
import torch.nn as nn

__all__ = ["ResNet", "BasicBlock", "Bottleneck", "resnet18", "resnet34", "resnet50", "resnet101", "resnet152",
           "model_urls"]

# torchvision's published ImageNet checkpoints (same table the reference downloads from)
model_urls = {
    "resnet18": "https://download.pytorch.org/models/resnet18-f37072fd.pth",
    "resnet34": "https://download.pytorch.org/models/resnet34-b627a593.pth",
    "resnet50": "https://download.pytorch.org/models/resnet50-0676ba61.pth",
    "resnet101": "https://download.pytorch.org/models/resnet101-63fe2227.pth",
    "resnet152": "https://download.pytorch.org/models/resnet152-394f9c45.pth",
}


def _conv(cin: int, cout: int, k: int, stride: int) -> nn.Conv2d:
    return nn.Conv2d(cin, cout, kernel_size=k, stride=stride, padding=k // 2, bias=False)


class _Residual(nn.Module):
    """Parameter container of one residual block.  ``plan`` lists (kernel, width-multiplier) of the convs in
    the main branch; the stride sits on the first 3x3 conv (ResNet v1.5, as in the reference)."""

    expansion = 1
    plan: Sequence = ()

    def __init__(self, inplanes: int, planes: int, stride: int = 1, downsample: Optional[nn.Module] = None,
                 groups: int = 1, base_width: int = 64, dilation: int = 1, norm_layer=None) -> None:
        super().__init__()
        if groups != 1 or base_width != 64 or dilation != 1 or norm_layer not in (None, nn.BatchNorm2d):
            raise NotImplementedError("only the plain ResNet family used by MSF-WSI is supported")
        cin = inplanes
        strided = False
        n = len(self.plan)
        for i, (k, mult) in enumerate(self.plan, start=1):
            cout = planes * mult
            s = 1
            if k == 3 and not strided:
                s, strided = stride, True
            setattr(self, f"conv{i}", _conv(cin, cout, k, s))
            setattr(self, f"bn{i}", nn.BatchNorm2d(cout))
            if i == 2 and n == 2:
                pass
            if (n == 2 and i == 1) or (n == 3 and i == 3):
                # keep the reference's registration order so modules() / repr line up
                self.relu = nn.ReLU(inplace=True)
            cin = cout
        self.downsample = downsample
        self.stride = stride

    def main_branch(self):
        return [(getattr(self, f"conv{i}"), getattr(self, f"bn{i}")) for i in range(1, len(self.plan) + 1)]

    def forward(self, x):  # pragma: no cover - blocks are executed by the engine as part of the encoder
        raise RuntimeError("residual blocks run inside the HIP engine; call the ResNet / MSFWSI module")


class BasicBlock(_Residual):
    expansion = 1
    plan = ((3, 1), (3, 1))


class Bottleneck(_Residual):
    expansion = 4
    plan = ((1, 1), (3, 1), (1, 4))


class ResNet(nn.Module):
    def __init__(self, block: Type[_Residual], layers: List[int], num_classes: int = 1000,
                 zero_init_residual: bool = False, groups: int = 1, width_per_group: int = 64,
                 replace_stride_with_dilation: Optional[List[bool]] = None, norm_layer=None,
                 return_features: bool = False) -> None:
        super().__init__()
        if groups != 1 or width_per_group != 64 or norm_layer not in (None, nn.BatchNorm2d):
            raise NotImplementedError("grouped / wide / custom-norm ResNets are outside the MSF-WSI hot path")
        if replace_stride_with_dilation not in (None, [False, False, False], (False, False, False)):
            raise NotImplementedError("dilated ResNets are outside the MSF-WSI hot path")
        self.return_features = return_features
        self.inplanes = 64
        self.conv1 = nn.Conv2d(3, 64, kernel_size=7, stride=2, padding=3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(kernel_size=3, stride=2, padding=1)
        for i, (planes, count) in enumerate(zip((64, 128, 256, 512), layers), start=1):
            setattr(self, f"layer{i}", self._stage(block, planes, count, stride=1 if i == 1 else 2))
        self.avgpool = nn.AdaptiveAvgPool2d((1, 1))
        self.fc = nn.Linear(512 * block.expansion, num_classes)

        # He-normal (fan_out) on every conv, BatchNorm affine = (1, 0); optionally zero the last BatchNorm
        # scale of each residual branch.  Module traversal order == RNG consumption order.
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.ones_(m.weight)
                nn.init.zeros_(m.bias)
        if zero_init_residual:
            for m in self.modules():
                if isinstance(m, _Residual):
                    nn.init.zeros_(m.main_branch()[-1][1].weight)
        to_kernel_layout_(self)

    def _stage(self, block, planes: int, count: int, stride: int) -> nn.Sequential:
        out_ch = planes * block.expansion
        shortcut = None
        if stride != 1 or self.inplanes != out_ch:
            shortcut = nn.Sequential(_conv(self.inplanes, out_ch, 1, stride), nn.BatchNorm2d(out_ch))
        blocks = [block(self.inplanes, planes, stride, shortcut)]
        self.inplanes = out_ch
        blocks += [block(out_ch, planes) for _ in range(1, count)]
        return nn.Sequential(*blocks)

    def stages(self):
        return [self.layer1, self.layer2, self.layer3, self.layer4]

    def forward(self, x: torch.Tensor):
        from .. import engine

        feats = engine.encoder_apply(self, x)
        last = feats[3]
        if not isinstance(self.fc, nn.Identity):  # classifier head is outside the hot path: plain torch
            last = self.fc(last.to(next(self.fc.parameters()).dtype))
        if self.return_features:
            return (feats[0], feats[1], feats[2], last)
        return last


def to_kernel_layout_(module: nn.Module) -> nn.Module:
    """Put every 4-D conv weight into channels_last storage (values and logical shape unchanged)."""
    for m in module.modules():
        if isinstance(m, nn.Conv2d) and m.weight.dim() == 4:
            m.weight.data = m.weight.data.contiguous(memory_format=torch.channels_last)
    return module


def _load_pretrained(arch: str, progress: bool):
    """ImageNet initialisation.  Same source as the reference (resnet.py:271-274); additionally an offline
    directory ``$MSFWSI_PRETRAINED_DIR/<file>.pth`` is honoured because the GPU boxes have no network."""
    url = model_urls[arch]
    local_dir = os.environ.get("MSFWSI_PRETRAINED_DIR")
    if local_dir:
        path = os.path.join(local_dir, os.path.basename(url))
        if os.path.isfile(path):
            return torch.load(path, map_location="cpu")
    return torch.hub.load_state_dict_from_url(url, progress=progress)


def _resnet(arch: str, block, layers, pretrained: bool, progress: bool, **kwargs: Any) -> ResNet:
    model = ResNet(block, layers, **kwargs)
    if pretrained:
        model.load_state_dict(_load_pretrained(arch, progress))
    return model


def resnet18(pretrained: bool = False, progress: bool = True, **kwargs: Any) -> ResNet:
    return _resnet("resnet18", BasicBlock, [2, 2, 2, 2], pretrained, progress, **kwargs)


def resnet34(pretrained: bool = False, progress: bool = True, **kwargs: Any) -> ResNet:
    return _resnet("resnet34", BasicBlock, [3, 4, 6, 3], pretrained, progress, **kwargs)


def resnet50(pretrained: bool = False, progress: bool = True, **kwargs: Any) -> ResNet:
    return _resnet("resnet50", Bottleneck, [3, 4, 6, 3], pretrained, progress, **kwargs)


def resnet101(pretrained: bool = False, progress: bool = True, **kwargs: Any) -> ResNet:
    return _resnet("resnet101", Bottleneck, [3, 4, 23, 3], pretrained, progress, **kwargs)


def resnet152(pretrained: bool = False, progress: bool = True, **kwargs: Any) -> ResNet:
    return _resnet("resnet152", Bottleneck, [3, 8, 36, 3], pretrained, progress, **kwargs)
