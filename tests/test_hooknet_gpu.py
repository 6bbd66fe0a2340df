"""Row f2 / BASELINE config 5: HookNet (two ResNet U-Nets + hook) forward and backward and the Dice loss on a real MI355X
against the torch-CPU oracle (oracle/hooknet_oracle.py, a restatement of smp's published algorithm -- parity unpinned) on
identical seeded weights and inputs, driven by the reference's own loop statements (tools/ssl_finetune.py:441-458)."""
import numpy as np
import pytest
import torch

from helpers import MODEL_SEED, rel, spread_gate

pytestmark = pytest.mark.gpu
CLASSES = 5  # + background channel -> 6 logit maps (the BCSS recipe: classes = len(class_names) + 1)


def _build(arch="resnet18"):
    from msf_wsi_amd.models.hooknet import HookNet

    torch.manual_seed(MODEL_SEED)
    return HookNet(encoder_name=arch, encoder_weights=None, classes=CLASSES + 1)


def _inputs(B=2, size=256, seed=3):
    g = torch.Generator().manual_seed(seed)
    x1, x2 = torch.randn(B, 3, size, size, generator=g), torch.randn(B, 3, size, size, generator=g)
    m1 = torch.randint(0, CLASSES + 1, (B, size, size), generator=g)
    m2 = torch.randint(0, CLASSES + 1, (B, size, size), generator=g)
    m2[m2 == 4] = 0  # a class that never occurs in the target masks: its Dice term is masked out
    return x1, x2, m1, m2


def _oracle(sd0, inputs, dt, lam=0.75):
    from oracle import hooknet_oracle as ho
    from oracle import msfwsi_oracle as orc

    x1, x2, m1, m2 = inputs
    sd = {k: (v.clone().to(dt) if v.is_floating_point() else v.clone()) for k, v in sd0.items()}
    params = {k: v for k, v in sd.items() if orc.is_param(k)}
    for v in params.values():
        v.requires_grad_(True)
    loss, (c, t) = ho.finetune_loss(sd, x1.to(dt), x2.to(dt), m1, m2, list(range(1, CLASSES + 1)), lam)
    loss.backward()
    return float(loss), c.detach(), t.detach(), {k: v.grad for k, v in params.items()}, sd


def test_state_dict_keys_follow_smp_naming(hip_lib):
    sd = _build().state_dict()
    for k in ("context_branch.encoder.conv1.weight", "context_branch.encoder.layer4.1.bn2.running_var",
              "context_branch.decoder.blocks.0.conv1.0.weight", "context_branch.decoder.blocks.4.conv2.1.num_batches_tracked",
              "context_branch.segmentation_head.0.bias", "target_branch.decoder.blocks.0.conv1.0.weight"):
        assert k in sd, k
    assert not any(".fc." in k for k in sd)
    assert tuple(sd["context_branch.decoder.blocks.0.conv1.0.weight"].shape) == (256, 512 + 256, 3, 3)
    assert tuple(sd["target_branch.decoder.blocks.0.conv1.0.weight"].shape) == (256, 512 + 128 + 256, 3, 3)  # + the hook
    assert tuple(sd["target_branch.segmentation_head.0.weight"].shape) == (CLASSES + 1, 16, 3, 3)
    # the encoder keys are what ssl_finetune.py:153-170 loads (strict) from a pre-train checkpoint
    from msf_wsi_amd.models import resnet

    enc = resnet.resnet18()
    want = {k for k in enc.state_dict() if not k.startswith("fc.")}
    got = {k[len("context_branch.encoder."):] for k in sd if k.startswith("context_branch.encoder.")}
    assert got == want


def test_hooknet_step_matches_oracle_fp32(hip_lib):
    from msf_wsi_amd.losses import MULTICLASS_MODE, DiceLoss

    model = _build()
    sd0 = {k: v.detach().clone() for k, v in model.state_dict().items()}
    inputs = _inputs()
    l64, c64, t64, g64, sd64 = _oracle(sd0, inputs, torch.float64)
    l32, c32, t32, g32, _ = _oracle(sd0, inputs, torch.float32)
    model = model.cuda().train()
    x1, x2, m1, m2 = [t.cuda() for t in inputs]
    lam = 0.75
    criterion = DiceLoss(MULTICLASS_MODE, classes=list(range(1, CLASSES + 1)), from_logits=True)
    # the reference loop's statements (tools/ssl_finetune.py:441-458)
    context_logits_mask, target_logits_mask = model(x1, x2)
    loss = (1 - lam) * criterion(context_logits_mask, m1) + lam * criterion(target_logits_mask, m2)
    loss.backward()
    torch.cuda.synchronize()
    assert tuple(target_logits_mask.shape) == (2, CLASSES + 1, 256, 256)
    assert rel(context_logits_mask, c64) < 1e-3 and rel(target_logits_mask, t64) < 1e-3
    assert abs(float(loss) - l64) <= 1e-3 * max(abs(l64), 1e-2), (float(loss), l64)
    named = list(model.named_parameters())
    names = [n for n, _ in named]
    assert all(p.grad is not None for _, p in named)
    rels = np.array([rel(p.grad, g64[n]) for n, p in named])
    box = np.array([rel(g32[n], g64[n]) for n in names])
    spread_gate(rels, names, [box], "HookNet gradients vs fp64 oracle")
    # BatchNorm running statistics of encoder and decoder moved exactly once
    now = model.state_dict()
    for k, v in sd64.items():
        if k.endswith("running_var"):
            assert torch.allclose(now[k].cpu().double(), v, rtol=1e-3, atol=1e-6), k
        if k.endswith("num_batches_tracked"):
            assert int(now[k]) == 1
    # an optimizer step in the reference's way (torch Adam on model.parameters(), ssl_finetune.py:289)
    torch.optim.Adam(model.parameters(), 1e-4).step()


def test_dice_kernel_matches_oracle(hip_lib):
    from msf_wsi_amd.losses import MULTICLASS_MODE, DiceLoss
    from oracle import hooknet_oracle as ho

    g = torch.Generator().manual_seed(8)
    logits = torch.randn(3, 6, 40, 36, generator=g) * 2
    target = torch.randint(0, 6, (3, 40, 36), generator=g)
    target[target == 2] = 0
    for classes in ([1, 2, 3, 4, 5], None):
        ref_in = logits.double().requires_grad_(True)
        ref = ho.dice_loss(ref_in, target, classes)
        ref.backward()
        x = logits.cuda().requires_grad_(True)
        out = DiceLoss(MULTICLASS_MODE, classes=classes, from_logits=True)(x, target.cuda())
        (out * 3.0).backward()
        torch.cuda.synchronize()
        assert abs(float(out) - float(ref)) < 1e-6
        assert rel(x.grad, 3.0 * ref_in.grad) < 1e-5


@pytest.mark.parametrize("arch", ["resnet18", "resnet50"])
def test_hooknet_bf16_autocast_and_eval(hip_lib, arch):
    """--amp path (bf16 storage / MFMA) with GradScaler, ResNet-18 and the Bottleneck family; eval-mode inference as in
    validate() (ssl_finetune.py:500-512)"""
    from msf_wsi_amd.losses import MULTICLASS_MODE, DiceLoss

    model = _build(arch).cuda().train()
    x1, x2, m1, m2 = [t.cuda() for t in _inputs(B=2)]
    criterion = DiceLoss(MULTICLASS_MODE, classes=list(range(1, CLASSES + 1)), from_logits=True)
    opt = torch.optim.Adam(model.parameters(), 1e-4)
    scaler = torch.amp.GradScaler("cuda", enabled=True)
    losses = []
    for _ in range(3):
        with torch.autocast("cuda", enabled=True, dtype=torch.bfloat16):
            c, t = model(x1, x2)
            loss = 0.25 * criterion(c, m1) + 0.75 * criterion(t, m2)
        opt.zero_grad()
        scaler.scale(loss).backward()
        scaler.step(opt)
        scaler.update()
        losses.append(float(loss))
    assert all(np.isfinite(losses)) and losses[-1] < losses[0]
    model.eval()
    with torch.no_grad():
        _, t = model(x1, x2)
    assert torch.isfinite(t).all() and tuple(t.shape) == (2, CLASSES + 1, 256, 256)


@pytest.mark.parametrize("lam", [1.0, 0.75])
def test_fused_finetune_step_matches_oracle(hip_lib, lam):
    """msf_wsi_amd.finetune.FinetuneStep (Dice kernels on the NHWC logits, flat Adam, device GradScaler) in fp32 against
    the oracle's finetune_loss + Adam on the same seeded weights and inputs: loss 1e-3, every updated weight tensor by
    the rule of the pre-train tests (Adam's first step is sign-like: helpers.updated_weights_gate).  lam = 1 is the
    reference's default (ssl_finetune.py:690): the context logits get no direct loss, only the hook's gradient.
    parity unpinned: smp absent (see oracle/hooknet_oracle.py)"""
    from helpers import updated_weights_gate
    from msf_wsi_amd.finetune import FinetuneStep
    from oracle import msfwsi_oracle as orc

    B = 4
    model = _build()
    sd0 = {k: v.detach().clone() for k, v in model.state_dict().items()}
    inputs = _inputs(B=B)
    x1, x2, m1, m2 = inputs
    lr = 1e-3 * (B ** 0.5) / (64 ** 0.5)
    res = {}
    for tag, dt in (("64", torch.float64), ("32", torch.float32)):
        loss, c, t, grads, sd = _oracle(sd0, inputs, dt, lam=lam)
        names = [k for k in sd if orc.is_param(k)]
        for k in names:
            sd[k].requires_grad_(False)
        opt = _OneGroupAdam(sd, names, lr)
        opt.step(sd, grads)
        res[tag] = (loss, grads, sd)
    loss64, g64, sd64 = res["64"]
    _, g32, sd32 = res["32"]
    model = model.cuda().train()
    ts = FinetuneStep(model, lr=1e-3, batch_size=B, lam=lam, dtype=torch.float32, use_scaler=False)
    loss, (tp, fp, fn, tn) = ts.step((x1.cuda(), x2.cuda()), (m1.cuda(), m2.cuda()))
    torch.cuda.synchronize()
    assert abs(float(loss) - loss64) <= 1e-3 * max(abs(loss64), 1e-2), (float(loss), loss64)
    named = list(model.named_parameters())
    names = [n for n, _ in named]
    box_step = np.array([rel(sd32[n], sd64[n]) for n in names])
    box_grad = np.array([rel(g32[n], g64[n]) if g64[n] is not None and float(g64[n].norm()) > 0 else 0.0 for n in names])
    g64z = {n: (g64[n] if g64[n] is not None else torch.zeros_like(sd0[n], dtype=torch.float64)) for n in names}
    updated_weights_gate(named, sd0, sd64, g64z, lr, [box_step], [box_grad], f"fused fine-tune step, lam {lam}")
    # the per-step confusion counts of the target prediction (ssl_finetune.py:440-447) against the oracle's logits
    from oracle import hooknet_oracle as ho
    from oracle import metrics_oracle as mo

    _, (c0_, t0_) = ho.finetune_loss({k: v.double() if v.is_floating_point() else v for k, v in sd0.items()},
                                      x1.double(), x2.double(), m1, m2, list(range(1, CLASSES + 1)), lam)
    want = mo.get_stats_multiclass(t0_.argmax(1).numpy() - 1, m2.numpy() - 1, CLASSES, ignore_index=-1)
    got = [v.cpu().numpy() for v in (tp, fp, fn, tn)]
    # an argmax can differ where two logits tie to rounding: allow a handful of pixels out of B * 65536
    assert sum(int(np.abs(a - b).sum()) for a, b in zip(got, want)) <= 64
    assert ts.t == 1 and int(model.state_dict()["context_branch.encoder.bn1.num_batches_tracked"]) == 1


class _OneGroupAdam:
    """torch.optim.Adam(model.parameters(), lr) restated on a state dict (one group: ssl_finetune.py:289)"""

    def __init__(self, sd, names, lr, eps=1e-8):
        self.names, self.lr, self.eps, self.t = names, lr, eps, 0
        self.m = {k: torch.zeros_like(sd[k]) for k in names}
        self.v = {k: torch.zeros_like(sd[k]) for k in names}

    @torch.no_grad()
    def step(self, sd, grads):
        import math

        self.t += 1
        b1, b2 = 0.9, 0.999
        bc1, bc2 = 1 - b1 ** self.t, 1 - b2 ** self.t
        for k in self.names:
            g = grads.get(k)
            if g is None:
                continue
            self.m[k].lerp_(g, 1 - b1)
            self.v[k].mul_(b2).addcmul_(g, g, value=1 - b2)
            sd[k].addcdiv_(self.m[k], (self.v[k].sqrt() / math.sqrt(bc2)).add_(self.eps), value=-(self.lr / bc1))


def test_fused_finetune_step_bf16_and_validate(hip_lib):
    """bf16 storage with the GradScaler protocol: finite, the loss tracks the fp32 oracle's to bf16 accuracy, the 16-bit
    weight copies follow the master weights; then the evaluation loop's arithmetic (chunks, eval-mode BatchNorm)"""
    from msf_wsi_amd.finetune import FinetuneStep

    B = 4
    model = _build()
    sd0 = {k: v.detach().clone() for k, v in model.state_dict().items()}
    x1, x2, m1, m2 = _inputs(B=B)
    loss32, _, _, _, _ = _oracle(sd0, (x1, x2, m1, m2), torch.float32, lam=1.0)
    model = model.cuda().train()
    ts = FinetuneStep(model, lr=1e-3, batch_size=B, dtype=torch.bfloat16)
    losses = [float(ts.step((x1.cuda(), x2.cuda()), (m1.cuda(), m2.cuda()))[0]) for _ in range(3)]
    torch.cuda.synchronize()
    assert all(np.isfinite(losses)) and abs(losses[0] - loss32) <= 2e-2 * abs(loss32), (losses, loss32)
    assert losses[2] < losses[0]  # the same batch three times: the loss falls
    assert ts.found_inf.item() == 0 and ts.scale.item() == 65536.0 and ts.t == 3
    assert torch.equal(ts.flats.w16[0].float(), ts.flats.w[0].bfloat16().float())
    tp, fp, fn, tn = ts.validate(x1.cuda().repeat(3, 1, 1, 1), x2.cuda().repeat(3, 1, 1, 1), m2.cuda().repeat(3, 1, 1),
                                 chunk=5)
    torch.cuda.synchronize()
    assert tuple(tp.shape) == (12, CLASSES) and not model.training
    assert torch.equal(tp[:4], tp[4:8]) and torch.equal(fn[:4], fn[8:12])  # eval mode: chunking does not change a tile
    valid = (m2 > 0).sum(dim=(1, 2))
    assert torch.equal((tp + fn).sum(1).cpu()[:4], valid)  # every labelled pixel is a tp or a fn of its class
