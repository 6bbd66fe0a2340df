"""ORACLE — test infrastructure only.  CPU restatement of the reference's MSF-WSI pre-training step.

Nothing in the product (msf_wsi_amd/) may import this module; only tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg do, as the checker / reported baseline.

It restates, with plain functional PyTorch on the CPU, the arithmetic of
  * ResNet encoder forward with multi-scale pooled features      reference src/models/resnet.py:232-256
    (BasicBlock :66-82, Bottleneck :120-140, stem :234-237, train-mode BatchNorm incl. running stats)
  * MSFWSI.forward (4 encoder calls, jigsaw un-shuffle, heads, fuser)        src/models/backbone.py:129-222
  * projector / predictor MLPs                                             src/models/backbone.py:12-31
  * the loss of the training loop                                  tools/ssl_train.py:422,448-466
  * Adam with 3 name-prefixed parameter groups and the GradScaler protocol  tools/ssl_train.py:281-310,471-474
operating on a *state dict* with the reference's key names (so it needs no module classes of its own and is
structurally independent of both the reference and the product).  Backward uses torch autograd on the CPU.

Pinning: tests/golden/make_golden.py imports the real reference from /root/reference in the build container,
asserts this restatement reproduces its outputs/gradients/updated weights on seeded inputs, and commits
the resulting vectors under tests/golden/ (see tests/test_oracle.py).  Parity status: PINNED for ResNet-18;
ResNet-50 uses the derived oracle described in SURVEY.md §8(c) (reference trunk + reference head factories
+ reference forward, width list scaled by the block expansion).
"""
from __future__ import annotations

import math
from typing import Dict, List, Sequence, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
StateDict = Dict[str, Tensor]

BN_EPS = 1e-5
BN_MOMENTUM = 0.1
FUSER_WEIGHTS = (0.1, 0.4, 0.7, 1.0)  # tools/ssl_train.py:623-625 default


# --------------------------------------------------------------------------------------------------
# structure discovery from key names
# --------------------------------------------------------------------------------------------------
def encoder_layout(sd: StateDict, prefix: str) -> List[List[Tuple[int, bool]]]:
    """[[(n_convs_in_block, has_downsample), ...] per stage] read off the state-dict keys."""
    stages = []
    for s in range(1, 5):
        blocks = []
        b = 0
        while f"{prefix}layer{s}.{b}.conv1.weight" in sd:
            nconv = 3 if f"{prefix}layer{s}.{b}.conv3.weight" in sd else 2
            blocks.append((nconv, f"{prefix}layer{s}.{b}.downsample.0.weight" in sd))
            b += 1
        stages.append(blocks)
    return stages


def _bn(sd: StateDict, key: str, x: Tensor, train: bool = True) -> Tensor:
    """train-mode batch norm with running-stat side effects, any of BatchNorm1d/2d (affine optional)"""
    w = sd.get(key + ".weight")
    b = sd.get(key + ".bias")
    out = F.batch_norm(x, sd[key + ".running_mean"], sd[key + ".running_var"], w, b, training=train,
                       momentum=BN_MOMENTUM, eps=BN_EPS)
    if train:
        sd[key + ".num_batches_tracked"] += 1
    return out


def encoder_forward(sd: StateDict, prefix: str, x: Tensor, train: bool = True) -> Tuple[Tensor, Tensor, Tensor, Tensor]:
    """resnet.py:232-256 with return_features=True and fc = Identity (backbone.py:64-65); train=False is the module
    under .eval(): every BatchNorm normalises with its running statistics"""
    bn = _bn
    if not train:
        def _bn_eval(sd_, key, x_):
            return bn(sd_, key, x_, train=False)
        return _encoder_forward(sd, prefix, x, _bn_eval)
    return _encoder_forward(sd, prefix, x, _bn)


def _encoder_forward(sd: StateDict, prefix: str, x: Tensor, _bn) -> Tuple[Tensor, Tensor, Tensor, Tensor]:
    y = F.conv2d(x, sd[prefix + "conv1.weight"], None, stride=2, padding=3)
    y = F.relu(_bn(sd, prefix + "bn1", y))
    y = F.max_pool2d(y, kernel_size=3, stride=2, padding=1)
    feats = []
    for s, blocks in enumerate(encoder_layout(sd, prefix), start=1):
        for b, (nconv, has_ds) in enumerate(blocks):
            p = f"{prefix}layer{s}.{b}."
            stride = 2 if (s > 1 and b == 0) else 1
            identity = y
            if nconv == 2:  # BasicBlock, resnet.py:66-82
                out = F.conv2d(y, sd[p + "conv1.weight"], None, stride=stride, padding=1)
                out = F.relu(_bn(sd, p + "bn1", out))
                out = F.conv2d(out, sd[p + "conv2.weight"], None, stride=1, padding=1)
                out = _bn(sd, p + "bn2", out)
            else:  # Bottleneck (stride on the 3x3), resnet.py:120-140
                out = F.conv2d(y, sd[p + "conv1.weight"], None)
                out = F.relu(_bn(sd, p + "bn1", out))
                out = F.conv2d(out, sd[p + "conv2.weight"], None, stride=stride, padding=1)
                out = F.relu(_bn(sd, p + "bn2", out))
                out = F.conv2d(out, sd[p + "conv3.weight"], None)
                out = _bn(sd, p + "bn3", out)
            if has_ds:
                identity = F.conv2d(y, sd[p + "downsample.0.weight"], None, stride=stride)
                identity = _bn(sd, p + "downsample.1", identity)
            y = F.relu(out + identity)
        feats.append(torch.flatten(F.adaptive_avg_pool2d(y, (1, 1)), 1))
    return tuple(feats)


def projector(sd: StateDict, key: str, x: Tensor) -> Tensor:
    """backbone.py:12-22 (Sequential indices 0..7)"""
    x = F.relu(_bn(sd, key + ".1", F.linear(x, sd[key + ".0.weight"])))
    x = F.relu(_bn(sd, key + ".4", F.linear(x, sd[key + ".3.weight"])))
    return _bn(sd, key + ".7", F.linear(x, sd[key + ".6.weight"]))


def predictor(sd: StateDict, key: str, x: Tensor) -> Tensor:
    """backbone.py:25-31"""
    x = F.relu(_bn(sd, key + ".1", F.linear(x, sd[key + ".0.weight"])))
    return F.linear(x, sd[key + ".3.weight"], sd[key + ".3.bias"])


def msfwsi_forward(sd: StateDict, x1, x2, jigsaw_idx, scale: int = 4, mask_ratio: float = 0.5, prefix: str = ""):
    """backbone.py:129-222.  Order of encoder calls (and hence of BatchNorm running-stat updates):
    context(view1), context(view2), target(view1), target(view2)."""
    K = int(scale ** 2)
    n_keep = int(K * (1 - mask_ratio))
    B = x1[0].shape[0]
    cf1 = encoder_forward(sd, prefix + "context_encoder.", x1[0])
    cf2 = encoder_forward(sd, prefix + "context_encoder.", x2[0])
    tf1 = encoder_forward(sd, prefix + "target_encoder.", x1[1])
    tf2 = encoder_forward(sd, prefix + "target_encoder.", x2[1])
    tf1s = [t.reshape(B, K, -1) for t in tf1]
    tf2s = [t.reshape(B, K, -1) for t in tf2]
    bidx = torch.arange(B, device=jigsaw_idx[0].device).repeat(K, 1).t()
    assert bidx.shape == jigsaw_idx[0].shape == jigsaw_idx[1].shape
    t1 = [t[bidx, jigsaw_idx[0], :].flatten(0, 1) for t in tf1s]
    t2 = [t[bidx, jigsaw_idx[1], :].flatten(0, 1) for t in tf2s]

    def heads(group: str, f1: Sequence[Tensor], f2: Sequence[Tensor]):
        z1 = [projector(sd, f"{prefix}{group}_projector.{i}", f) for i, f in enumerate(f1)]
        z2 = [projector(sd, f"{prefix}{group}_projector.{i}", f) for i, f in enumerate(f2)]
        p1 = [predictor(sd, f"{prefix}{group}_predictor.{i}", z) for i, z in enumerate(z1)]
        p2 = [predictor(sd, f"{prefix}{group}_predictor.{i}", z) for i, z in enumerate(z2)]
        return (tuple(p1), tuple(p2), tuple(z.detach() for z in z1), tuple(z.detach() for z in z2))

    ctx = heads("context", cf1, cf2)
    tgt = heads("target", t1, t2)
    ms1 = [torch.cat((c, t[:, :n_keep, :].flatten(1)), dim=1) for c, t in zip(cf1, tf1s)]
    ms2 = [torch.cat((c, t[:, :n_keep, :].flatten(1)), dim=1) for c, t in zip(cf2, tf2s)]
    # the reference interleaves per scale: proj(v1), proj(v2), pred(v1), pred(v2) (backbone.py:205-212)
    z1, z2, p1, p2 = [], [], [], []
    for i in range(4):
        z1.append(projector(sd, f"{prefix}inter_projector.{i}", ms1[i]))
        z2.append(projector(sd, f"{prefix}inter_projector.{i}", ms2[i]))
        p1.append(predictor(sd, f"{prefix}inter_predictor.{i}", z1[i]))
        p2.append(predictor(sd, f"{prefix}inter_predictor.{i}", z2[i]))
    ms = (tuple(p1), tuple(p2), tuple(z.detach() for z in z1), tuple(z.detach() for z in z2))
    return ctx, tgt, ms


def loss_terms(outputs, weights: Sequence[float] = FUSER_WEIGHTS) -> Tuple[Tensor, List[List[Tensor]]]:
    """tools/ssl_train.py:448-466: sum over 3 groups x 4 scales of w_s * -(cos(p1,z2).mean()+cos(p2,z1).mean())/2"""
    total = 0
    terms = []
    for grp in outputs:
        row = []
        for i, (p1, p2, z1, z2) in enumerate(zip(*grp)):
            if p1.dtype in (torch.float16, torch.bfloat16):  # autocast runs cosine_similarity in fp32
                p1, p2, z1, z2 = p1.float(), p2.float(), z1.float(), z2.float()
            t = -(F.cosine_similarity(p1, z2, dim=1).mean() + F.cosine_similarity(p2, z1, dim=1).mean()) * 0.5
            row.append(t.detach())
            total = total + t * weights[i]
        terms.append(row)
    return total, terms


def infonce_terms(outputs, weights: Sequence[float] = FUSER_WEIGHTS, temperature: float = 0.2, gathered=None):
    """The InfoNCE variant named by BASELINE.json's north_star -- NOT in the reference (its loss is `loss_terms` above;
    SURVEY D1), hence PARITY UNPINNED: a plain torch statement of the standard formulation, used only to check the
    product's optional mode.  Per (p, z) pair: cross entropy of normalize(p) @ normalize(z_all)^T / tau against the own
    row, z_all = the z rows of all ranks (`gathered(z)`; identity for one process); weights and the 1/2 as in loss_terms."""
    total = 0
    terms = []
    for grp in outputs:
        row = []
        for i, (p1, p2, z1, z2) in enumerate(zip(*grp)):
            t = 0
            for p, z in ((p1, z2), (p2, z1)):
                z_all, off = (z, 0) if gathered is None else gathered(z)
                logits = F.normalize(p.float(), dim=1) @ F.normalize(z_all.float(), dim=1).t() / temperature
                t = t + 0.5 * F.cross_entropy(logits, torch.arange(p.shape[0]) + off)
            row.append(t.detach())
            total = total + t * weights[i]
        terms.append(row)
    return total, terms


# --------------------------------------------------------------------------------------------------
# parameters / optimizer
# --------------------------------------------------------------------------------------------------
def is_param(key: str) -> bool:
    return not key.endswith(("running_mean", "running_var", "num_batches_tracked"))


def param_groups(sd: StateDict) -> List[List[str]]:
    """tools/ssl_train.py:281-300: groups by name prefix in named_parameters() order (= state-dict order)"""
    keys = [k for k in sd if is_param(k)]
    return [[k for k in keys if k.startswith(p)] for p in ("context_", "target_", "inter_")]


def init_lr(lr: float, global_batch: int) -> float:
    return lr * math.sqrt(global_batch) / math.sqrt(32)  # tools/ssl_train.py:155


class Adam:
    """torch.optim.Adam defaults (betas .9/.999, eps 1e-8, no weight decay), restated explicitly."""

    def __init__(self, sd: StateDict, lrs: Sequence[float], eps: float = 1e-8):
        self.groups = param_groups(sd)
        self.lrs = list(lrs)
        self.eps = eps
        self.t = 0
        self.m = {k: torch.zeros_like(sd[k]) for g in self.groups for k in g}
        self.v = {k: torch.zeros_like(sd[k]) for g in self.groups for k in g}

    @torch.no_grad()
    def step(self, sd: StateDict, grads: Dict[str, Tensor]):
        self.t += 1
        b1, b2 = 0.9, 0.999
        bc1, bc2 = 1 - b1 ** self.t, 1 - b2 ** self.t
        for g, lr in zip(self.groups, self.lrs):
            for k in g:
                if k not in grads or grads[k] is None:
                    continue
                gr = grads[k]
                self.m[k].lerp_(gr, 1 - b1)
                self.v[k].mul_(b2).addcmul_(gr, gr, value=1 - b2)
                denom = (self.v[k].sqrt() / math.sqrt(bc2)).add_(self.eps)
                sd[k].addcdiv_(self.m[k], denom, value=-(lr / bc1))


def train_step(sd: StateDict, batch, opt: Adam, scale: int = 4, mask_ratio: float = 0.5,
               weights: Sequence[float] = FUSER_WEIGHTS, loss_scale: float = 1.0, loss_fn=None, autocast_dtype=None):
    """one iteration of tools/ssl_train.py:425-474.  autocast_dtype=None: plain fp32 / fp64 arithmetic of the state
    dict's dtype; torch.bfloat16 / torch.float16: forward and loss under `torch.autocast("cpu", dtype)` as the
    reference loop runs them under --amp (ssl_train.py:441; master weights, gradients and Adam stay fp32) -- the
    yardstick of the product's 16-bit runs.  Returns loss, per-term losses, the forward outputs and the (unscaled)
    gradients; updates `sd` in place."""
    (c1, c2), (t1, t2), idx = batch
    params = {k: v for k, v in sd.items() if is_param(k)}
    for v in params.values():
        v.requires_grad_(True)
        v.grad = None
    if autocast_dtype is None:
        outputs = msfwsi_forward(sd, (c1, t1), (c2, t2), idx, scale, mask_ratio)
        loss, terms = (loss_fn or loss_terms)(outputs, weights)
    else:
        with torch.autocast(c1.device.type, dtype=autocast_dtype):  # "cpu" here; "cuda" for bench.py's stock-GPU yardstick
            outputs = msfwsi_forward(sd, (c1, t1), (c2, t2), idx, scale, mask_ratio)
            loss, terms = (loss_fn or loss_terms)(outputs, weights)
    (loss * loss_scale).backward()
    grads = {}
    for k, v in params.items():
        v.requires_grad_(False)
        grads[k] = None if v.grad is None else (v.grad if loss_scale == 1.0 else v.grad / loss_scale)
        v.grad = None
    finite = all(torch.isfinite(g).all() for g in grads.values() if g is not None)
    if finite:
        opt.step(sd, grads)
    return loss.detach(), terms, outputs, grads


def diverse_batch(B: int, size: int = 64, K: int = 16, seed: int = 0, dtype=torch.float32):
    """Well-conditioned synthetic input with the same batch contract as `synthetic_batch`: every image is a smooth
    random pattern (two scales of block-constant noise + mild pixel noise) with its OWN mean and contrast, as stained
    tissue tiles differ from each other.  With i.i.d. N(0,1) pixels all pooled features of a batch are nearly equal,
    the heads' BatchNorm1d divides by a vanishing batch deviation and the reference's own fp32 run sits 2e-2 from its
    fp64 run; on these inputs the reference's fp32<->fp64 spread has a median of ~1e-5.
    Only exactly reproducible operations (randn, repeat_interleave, one multiply, one add per element): the same
    seed gives bit-identical tensors on every machine.  size must be a multiple of 8."""
    g = torch.Generator().manual_seed(seed)
    c1, c2, t1, t2 = (_diverse_images(g, n, size) for n in (B, B, B * K, B * K))
    i1 = torch.stack([torch.argsort(torch.randperm(K, generator=g)) for _ in range(B)])
    i2 = torch.stack([torch.argsort(torch.randperm(K, generator=g)) for _ in range(B)])
    return (c1.to(dtype), c2.to(dtype)), (t1.to(dtype), t2.to(dtype)), [i1, i2]


def _diverse_images(g: torch.Generator, n: int, size: int) -> Tensor:
    if size % 8:
        raise ValueError("diverse inputs: size must be a multiple of 8")
    coarse = torch.randn(n, 3, 8, 8, generator=g).repeat_interleave(size // 8, 2).repeat_interleave(size // 8, 3)
    fine = torch.randn(n, 3, size // 2, size // 2, generator=g).repeat_interleave(2, 2).repeat_interleave(2, 3)
    x = coarse + 0.5 * fine
    x = x + 0.25 * torch.randn(n, 3, size, size, generator=g)
    mean = 0.8 * torch.randn(n, 3, 1, 1, generator=g)
    contrast = torch.exp(0.5 * torch.randn(n, 1, 1, 1, generator=g))
    return x * contrast + mean


def diverse_images(n: int, size: int = 64, seed: int = 0, dtype=torch.float32) -> Tensor:
    """n images of the `diverse_batch` kind (encoder-only tests)"""
    return _diverse_images(torch.Generator().manual_seed(seed), n, size).to(dtype)


def make_batch(kind: str, B: int, size: int, K: int = 16, seed: int = 0, dtype=torch.float32):
    """kind = "normal" (SURVEY 8(d) inputs) | "diverse" (well-conditioned parity inputs)"""
    return (diverse_batch if kind == "diverse" else synthetic_batch)(B, size, K, seed, dtype)


def synthetic_batch(B: int, size: int = 224, K: int = 16, seed: int = 0, dtype=torch.float32):
    """N(0,1) images in the order ctx0, ctx1, tgt0, tgt1, then the inverse jigsaw
    permutations idx0[0..B), idx1[0..B) (argsort(randperm(K)), src/utils/data/bcss.py:171-172)."""
    g = torch.Generator().manual_seed(seed)
    c1 = torch.randn(B, 3, size, size, generator=g)
    c2 = torch.randn(B, 3, size, size, generator=g)
    t1 = torch.randn(B * K, 3, size, size, generator=g)
    t2 = torch.randn(B * K, 3, size, size, generator=g)
    i1 = torch.stack([torch.argsort(torch.randperm(K, generator=g)) for _ in range(B)])
    i2 = torch.stack([torch.argsort(torch.randperm(K, generator=g)) for _ in range(B)])
    return (c1.to(dtype), c2.to(dtype)), (t1.to(dtype), t2.to(dtype)), [i1, i2]
