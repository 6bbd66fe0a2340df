import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import MODEL_SEED, load_golden, rel
from msf_wsi_amd.engine import Engine
from msf_wsi_amd.models import resnet
vec, man = load_golden("r50enc_b4_s64")
B, size = man["B"], man["size"]
for fwd in (False, True):
    torch.manual_seed(MODEL_SEED)
    enc = resnet.resnet50(zero_init_residual=False, return_features=True)
    enc.fc = torch.nn.Identity()
    enc = enc.cuda().train()
    enc._engine = Engine(); enc._engine.fold_bn3_fwd = fwd
    g = torch.Generator().manual_seed(man["data_seed"])
    x = torch.randn(B, 3, size, size, generator=g)
    Rs = [torch.randn(B, d, generator=g) for d in man["feature_dims"]]
    feats = enc(x.cuda())
    loss = sum((f * r.cuda()).sum() for f, r in zip(feats, Rs))
    loss.backward(); torch.cuda.synchronize()
    named = dict(enc.named_parameters())
    print("fold_fwd", fwd, "feat rel", [f"{rel(f, vec[f'feat/{s}']):.2e}" for s, f in enumerate(feats)])
    norms = np.array([float(named[k].grad.double().norm()) for k in man["param_keys"]])
    rn = np.abs(norms - vec["grad_norm"]) / (vec["grad_norm"] + 1e-30)
    print("   grad-norm rel: median %.2e max %.2e" % (np.median(rn), rn.max()))
    for k in ("conv1.weight", "layer1.0.downsample.1.weight", "layer2.0.bn2.bias", "layer4.2.bn3.weight"):
        if f"grad/{k}" in vec: print("   ", k, "%.3e" % rel(named[k].grad, vec[f"grad/{k}"]))
