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
