"""ORACLE -- test infrastructure only (nothing under msf_wsi_amd/ may import it).

Functional torch-CPU restatement, on a state dict with the reference's key names, of
  * HookNet.forward (two smp.Unet branches + hook)                       reference src/models/hooknet.py:15-35,84-100,246-252
  * smp.losses.DiceLoss(MULTICLASS_MODE, classes, from_logits=True)      reference tools/ssl_finetune.py:287-288,444-447
PARITY UNPINNED: the decoder / loss arithmetic lives in `segmentation-models-pytorch>=0.3.2`
(/root/reference/environment.yml:26; not pinned exactly, not under /root/reference, absent from this image); the reference
has no test for it.  Restated from the package's published code: ResNetEncoder stages [identity, conv1-bn1-relu,
maxpool-layer1, layer2, layer3, layer4]; UnetDecoder drops the first feature, reverses, and per DecoderBlock does
F.interpolate(scale_factor=2, mode="nearest") -> cat(skip) -> [conv3x3-BN-ReLU] x2; SegmentationHead = conv3x3;
DiceLoss: softmax, one-hot, per-class soft dice over (batch, pixels), (1 - dice) * [class present], mean over `classes`.
The encoder part is the pinned trunk of oracle/msfwsi_oracle.py."""
from __future__ import annotations

import torch
import torch.nn.functional as F

from . import msfwsi_oracle as orc


def encoder_features(sd, prefix, x):
    """[relu(bn1(conv1 x)), layer1, layer2, layer3, layer4] maps"""
    y = F.conv2d(x, sd[prefix + "conv1.weight"], None, stride=2, padding=3)
    a0 = F.relu(orc._bn(sd, prefix + "bn1", y))
    y = F.max_pool2d(a0, kernel_size=3, stride=2, padding=1)
    feats = [a0]
    for s, blocks in enumerate(orc.encoder_layout(sd, prefix), start=1):
        for b, (nconv, has_ds) in enumerate(blocks):
            p = f"{prefix}layer{s}.{b}."
            stride = 2 if (s > 1 and b == 0) else 1
            identity = y
            if nconv == 2:
                out = F.relu(orc._bn(sd, p + "bn1", F.conv2d(y, sd[p + "conv1.weight"], None, stride=stride, padding=1)))
                out = orc._bn(sd, p + "bn2", F.conv2d(out, sd[p + "conv2.weight"], None, stride=1, padding=1))
            else:
                out = F.relu(orc._bn(sd, p + "bn1", F.conv2d(y, sd[p + "conv1.weight"], None)))
                out = F.relu(orc._bn(sd, p + "bn2", F.conv2d(out, sd[p + "conv2.weight"], None, stride=stride, padding=1)))
                out = orc._bn(sd, p + "bn3", F.conv2d(out, sd[p + "conv3.weight"], None))
            if has_ds:
                identity = orc._bn(sd, p + "downsample.1", F.conv2d(y, sd[p + "downsample.0.weight"], None, stride=stride))
            y = F.relu(out + identity)
        feats.append(y)
    return feats


def decoder(sd, prefix, feats, context_feats=None, hook=False):
    feats = feats[::-1]
    head, skips = feats[0], feats[1:]
    if context_feats is not None:
        head = torch.cat([head, context_feats], dim=1)
    x, hooked = head, None
    for i in range(5):
        x = F.interpolate(x, scale_factor=2, mode="nearest")
        if i < len(skips):
            x = torch.cat([x, skips[i]], dim=1)
        for j in (1, 2):
            k = f"{prefix}blocks.{i}.conv{j}."
            x = F.relu(orc._bn(sd, k + "1", F.conv2d(x, sd[k + "0.weight"], None, padding=1)))
        if hook and i == 1:
            hooked = x[:, :, 16 - 4:16 + 4, 16 - 4:16 + 4]
    return x, hooked


def hooknet_forward(sd, x1, x2):
    cf = encoder_features(sd, "context_branch.encoder.", x1)
    cx, hooked = decoder(sd, "context_branch.decoder.", cf, hook=True)
    cmask = F.conv2d(cx, sd["context_branch.segmentation_head.0.weight"], sd["context_branch.segmentation_head.0.bias"],
                     padding=1)
    tf = encoder_features(sd, "target_branch.encoder.", x2)
    tx, _ = decoder(sd, "target_branch.decoder.", tf, context_feats=hooked)
    tmask = F.conv2d(tx, sd["target_branch.segmentation_head.0.weight"], sd["target_branch.segmentation_head.0.bias"],
                     padding=1)
    return cmask, tmask


def dice_loss(logits, target, classes, smooth=0.0, eps=1e-7):
    bs, C = logits.shape[0], logits.shape[1]
    p = logits.log_softmax(dim=1).exp().view(bs, C, -1)
    t = F.one_hot(target.view(bs, -1), C).permute(0, 2, 1).to(p.dtype)
    inter = (p * t).sum((0, 2))
    card = (p + t).sum((0, 2))
    dice = (2.0 * inter + smooth) / (card + smooth).clamp_min(eps)
    loss = (1.0 - dice) * (t.sum((0, 2)) > 0).to(p.dtype)
    if classes is not None:
        loss = loss[list(classes)]
    return loss.mean()


def finetune_loss(sd, x1, x2, m1, m2, classes, lam):
    """tools/ssl_finetune.py:441-447"""
    c, t = hooknet_forward(sd, x1, x2)
    return (1 - lam) * dice_loss(c, m1, classes) + lam * dice_loss(t, m2, classes), (c, t)
