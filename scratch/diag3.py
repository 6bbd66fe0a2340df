import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch, numpy as np
from helpers import *
from oracle import msfwsi_oracle as orc
torch.set_num_threads(int(os.environ.get("NT", "128")))
model = build_product("resnet18")
sd0 = {k: v.detach().clone() for k, v in model.state_dict().items()}
batch = orc.synthetic_batch(8, 64, 16, 0)
orig = orc._bn
def run(dt):
    store = {}
    def wrapped(sd, key, x, train=True):
        if x.requires_grad: x.retain_grad()
        out = orig(sd, key, x, train)
        out.retain_grad()
        store.setdefault(key, []).append((x, out))
        return out
    orc._bn = wrapped
    osd = {k: (v.to(dt) if v.is_floating_point() else v.clone()) for k, v in sd0.items()}
    (c1, c2), (t1, t2), idx = batch
    b = ((c1.to(dt), c2.to(dt)), (t1.to(dt), t2.to(dt)), idx)
    opt = orc.Adam(osd, [1e-3]*3); opt.step = lambda *a, **k: None
    r = orc.train_step(osd, b, opt, 4, 0.5, WEIGHTS)
    orc._bn = orig
    return r, store
(r32, s32), (r64, s64) = run(torch.float32), run(torch.float64)
for key in s32:
    if not (key.startswith("context_encoder.layer2") or key.startswith("context_encoder.layer3.0") or key.startswith("context_encoder.layer1.1") or key.startswith("target_projector.2") or key.startswith("target_predictor.2")): continue
    for v in range(2):
        (x32, o32), (x64, o64) = s32[key][v], s64[key][v]
        var = x64.detach().transpose(0,1).reshape(x64.shape[1], -1).var(1, unbiased=False)
        print("%-45s v%d  d_out %.2e  d_in %.2e  fwd_in %.2e  min var %.3e" % (key, v, rel(o32.grad, o64.grad), rel(x32.grad, x64.grad), rel(x32, x64), var.min().item()))
print("=========== detail target_projector.2.4 v1")
(x32, o32), (x64, o64) = s32["target_projector.2.4"][1], s64["target_projector.2.4"][1]
m32, m64 = (o32 > 0), (o64 > 0)
mis = (m32 != m64)
print("sign mismatches", int(mis.sum()), "exact zeros o32", int((o32 == 0).sum()), "of", o32.numel())
if mis.any():
    print("o64 at mismatches", o64[mis][:10].tolist(), "o32", o32[mis][:10].tolist())
d = (o32.grad.double() - o64.grad).abs()
print("grad diff: max %.3e  ref max %.3e  n(|d|>1e-3 max) %d" % (d.max().item(), o64.grad.abs().max().item(), int((d > 1e-3 * o64.grad.abs().max()).sum())))
rows = (d > 1e-3 * o64.grad.abs().max()).nonzero()
print("rows/cols of big diffs", rows[:12].tolist())
# per-row norm diffs
rd = (o32.grad.double() - o64.grad).norm(dim=1) / (o64.grad.norm(dim=1) + 1e-30)
print("rows with rel diff > 1e-3:", (rd > 1e-3).nonzero().flatten().tolist()[:20], "max", rd.max().item())
(x32b, o32b), (x64b, o64b) = s32["target_projector.2.7"][1], s64["target_projector.2.7"][1]
rd7 = (x32b.grad.double() - x64b.grad).norm(dim=1) / (x64b.grad.norm(dim=1) + 1e-30)
print(".7 d_in rows rel max", rd7.max().item())
