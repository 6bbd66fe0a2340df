"""Drop-in for the reference's ``src/models/hooknet.py`` (row f2 of SURVEY.md 8f, BASELINE config 5): the two-branch
"HookNet" U-Net that ``tools/ssl_finetune.py`` fine-tunes on top of the pre-trained encoders.

The reference builds it from ``segmentation_models_pytorch`` (``smp.Unet`` subclasses, hooknet.py:102-254) -- a third-party
package that is not part of the reference tree and is absent from this image.  This module therefore restates the
*module tree* smp would build for a ResNet encoder, with the same attribute names, so that state-dict keys line up with
checkpoints written by the reference (``context_branch.encoder.layer1.0.conv1.weight``,
``target_branch.decoder.blocks.0.conv1.0.weight``, ``context_branch.segmentation_head.0.bias`` ...) and the encoder keys
are exactly the ones ``ssl_finetune.py:153-170`` loads from a pre-train checkpoint (strict):

    HookNet(encoder_name, encoder_depth=5, encoder_weights, decoder_use_batchnorm=True,
            decoder_channels=(256,128,64,32,16), decoder_attention_type=None, in_channels=3, classes, activation,
            aux_params)                                                        hooknet.py:210-252
      .context_branch / .target_branch : encoder (torchvision-named ResNet, no fc), decoder.blocks[i].conv{1,2} =
        [Conv2d(3x3, bias=False), BatchNorm2d, ReLU], segmentation_head = [Conv2d(3x3), Identity, Identity]
      forward(x1, x2) -> (context_masks, target_masks)                          hooknet.py:248-252

Arithmetic (forward and backward of both branches: encoder with skip connections, nearest-x2 upsample + concat,
conv-BN-ReLU pairs, the 12:20 centre-crop hook, the segmentation heads) runs in the HIP kernels through
:mod:`msf_wsi_amd.unet_engine`.  Parity status: UNPINNED (third-party arithmetic; checked against
a torch restatement of smp's published algorithm).  Supported: ResNet encoders, depth 5, batch-norm decoders without
attention, no activation / aux head -- the configuration every reference script uses; anything else raises."""
from __future__ import annotations

from typing import List, Optional

import torch
import torch.nn as nn

from . import resnet as _resnet


class Conv2dReLU(nn.Sequential):
    """smp.base.modules.Conv2dReLU with use_batchnorm=True: [Conv2d(bias=False), BatchNorm2d, ReLU]"""

    def __init__(self, in_channels: int, out_channels: int, kernel_size: int = 3, padding: int = 1):
        super().__init__(nn.Conv2d(in_channels, out_channels, kernel_size, padding=padding, bias=False),
                         nn.BatchNorm2d(out_channels), nn.ReLU(inplace=True))


class _Attention(nn.Module):
    """smp.base.modules.Attention(name=None): an Identity holder (kept for the module tree / repr)"""

    def __init__(self):
        super().__init__()
        self.attention = nn.Identity()


class DecoderBlock(nn.Module):
    """smp unet DecoderBlock: nearest x2 -> cat(skip) -> conv1 -> conv2 (attention = Identity)"""

    def __init__(self, in_channels: int, skip_channels: int, out_channels: int):
        super().__init__()
        self.conv1 = Conv2dReLU(in_channels + skip_channels, out_channels)
        self.attention1 = _Attention()
        self.conv2 = Conv2dReLU(out_channels, out_channels)
        self.attention2 = _Attention()
        self.in_channels, self.skip_channels, self.out_channels = in_channels, skip_channels, out_channels


class UnetDecoder(nn.Module):
    """smp UnetDecoder for a depth-5 encoder; ``extra_head`` = the 128 hooked context channels concatenated to the
    target branch's head (hooknet.py:63-65, 'hardcoded' there)"""

    def __init__(self, encoder_channels, decoder_channels, extra_head: int = 0):
        super().__init__()
        if len(decoder_channels) != 5:
            raise ValueError("Model depth is 5, but you provide `decoder_channels` for "
                             f"{len(decoder_channels)} blocks.")
        enc = list(encoder_channels[1:])[::-1]
        head = enc[0]
        in_ch = [head + extra_head] + list(decoder_channels[:-1])
        skip_ch = list(enc[1:]) + [0]
        self.center = nn.Identity()
        self.blocks = nn.ModuleList([DecoderBlock(i, s, o) for i, s, o in zip(in_ch, skip_ch, decoder_channels)])


class SegmentationHead(nn.Sequential):
    """smp SegmentationHead(kernel_size=3, activation=None, upsampling=1): [Conv2d, Identity, Identity]"""

    def __init__(self, in_channels: int, out_channels: int):
        super().__init__(nn.Conv2d(in_channels, out_channels, kernel_size=3, padding=1), nn.Identity(), nn.Identity())


def _init_decoder(module: nn.Module):
    """smp.base.initialization.initialize_decoder"""
    for m in module.modules():
        if isinstance(m, nn.Conv2d):
            nn.init.kaiming_uniform_(m.weight, mode="fan_in", nonlinearity="relu")
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)
        elif isinstance(m, nn.BatchNorm2d):
            nn.init.constant_(m.weight, 1)
            nn.init.constant_(m.bias, 0)


def _init_head(module: nn.Module):
    """smp.base.initialization.initialize_head"""
    for m in module.modules():
        if isinstance(m, (nn.Linear, nn.Conv2d)):
            nn.init.xavier_uniform_(m.weight)
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)


_ENCODERS = {"resnet18": (_resnet.BasicBlock, [2, 2, 2, 2]), "resnet34": (_resnet.BasicBlock, [3, 4, 6, 3]),
             "resnet50": (_resnet.Bottleneck, [3, 4, 6, 3]), "resnet101": (_resnet.Bottleneck, [3, 4, 23, 3]),
             "resnet152": (_resnet.Bottleneck, [3, 8, 36, 3])}


def _encoder(name: str, weights: Optional[str]) -> nn.Module:
    if name not in _ENCODERS:
        raise NotImplementedError(f"encoder {name!r}: only the ResNet family of the MSF-WSI scripts is supported")
    block, layers = _ENCODERS[name]
    enc = _resnet.ResNet(block, layers)
    if weights is not None:  # smp's "imagenet" = the torchvision checkpoint the pre-training also starts from
        enc.load_state_dict(_resnet._load_pretrained(name, True))
    enc.fc = nn.Identity()  # smp's ResNetEncoder deletes fc: no fc.* keys in the state dict
    e = block.expansion
    enc.out_channels = (3, 64, 64 * e, 128 * e, 256 * e, 512 * e)
    return enc


class _Unet(nn.Module):
    def __init__(self, encoder_name, encoder_depth, encoder_weights, decoder_use_batchnorm, decoder_channels,
                 decoder_attention_type, in_channels, classes, activation, aux_params, extra_head: int):
        super().__init__()
        if encoder_depth != 5 or not decoder_use_batchnorm or decoder_attention_type is not None or in_channels != 3 \
                or activation is not None or aux_params is not None:
            raise NotImplementedError("HookNet on MI355X supports the configuration of the reference scripts: depth 5, "
                                      "batch-norm decoder, no attention, 3 input channels, no activation / aux head")
        self.encoder = _encoder(encoder_name, encoder_weights)
        self.decoder = UnetDecoder(self.encoder.out_channels, tuple(decoder_channels), extra_head)
        self.segmentation_head = SegmentationHead(decoder_channels[-1], classes)
        self.classification_head = None
        self.name = "u-{}".format(encoder_name)
        _init_decoder(self.decoder)
        _init_head(self.segmentation_head)
        _resnet.to_kernel_layout_(self)


class ContextUnet(_Unet):
    """hooknet.py:102-154: smp.Unet whose decoder also returns the hooked features"""

    def __init__(self, encoder_name="resnet18", encoder_depth=5, encoder_weights="imagenet", decoder_use_batchnorm=True,
                 decoder_channels=(256, 128, 64, 32, 16), decoder_attention_type=None, in_channels=3, classes=1,
                 activation=None, aux_params=None):
        super().__init__(encoder_name, encoder_depth, encoder_weights, decoder_use_batchnorm, decoder_channels,
                         decoder_attention_type, in_channels, classes, activation, aux_params, extra_head=0)


class TargetUnet(_Unet):
    """hooknet.py:157-207: smp.Unet whose decoder head takes the 128 hooked context channels as well"""

    def __init__(self, encoder_name="resnet18", encoder_depth=5, encoder_weights="imagenet", decoder_use_batchnorm=True,
                 decoder_channels=(256, 128, 64, 32, 16), decoder_attention_type=None, in_channels=3, classes=1,
                 activation=None, aux_params=None):
        super().__init__(encoder_name, encoder_depth, encoder_weights, decoder_use_batchnorm, decoder_channels,
                         decoder_attention_type, in_channels, classes, activation, aux_params, extra_head=128)


class HookNet(nn.Module):
    """hooknet.py:210-254.  ``forward(x1, x2)``: x1 = context images, x2 = target images, both [N,3,H,W] (the reference
    feeds 256x256: the hook crop 12:20 is hard-coded for that size); returns (context_masks, target_masks) logits
    [N, classes, H, W] (fp32), autograd-connected to every parameter."""

    def __init__(self, encoder_name: str = "resnet18", encoder_depth: int = 5, encoder_weights: Optional[str] = "imagenet",
                 decoder_use_batchnorm: bool = True, decoder_channels: List[int] = (256, 128, 64, 32, 16),
                 decoder_attention_type: Optional[str] = None, in_channels: int = 3, classes: int = 1,
                 activation=None, aux_params: Optional[dict] = None):
        super().__init__()
        args = (encoder_name, encoder_depth, encoder_weights, decoder_use_batchnorm, decoder_channels,
                decoder_attention_type, in_channels, classes, activation, aux_params)
        self.encoder_name = encoder_name  # the fine-tune checkpoint's "arch" entry (ssl_finetune.py:355)
        self.context_branch = ContextUnet(*args)
        self.target_branch = TargetUnet(*args)

    def forward(self, x1: torch.Tensor, x2: torch.Tensor):
        from .. import unet_engine

        return unet_engine.hooknet_apply(self, x1, x2)
