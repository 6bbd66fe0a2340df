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
orig = orc.encoder_forward
def run(dt):
    store = []
    def wrapped(sd, prefix, x):
        f = orig(sd, prefix, x)
        for t in f: t.retain_grad()
        store.append(f)
        return f
    orc.encoder_forward = wrapped
    osd = {k: (v.to(dt) if v.is_floating_point() else v.clone()) for k, v in sd0.items()}
    (c1, c2), (t1, t2), idx = batch
    b = ((c1.to(dt), c2.to(dt)), (t1.to(dt), t2.to(dt)), idx)
    opt = orc.Adam(osd, [1e-3]*3); opt.step = lambda *a, **k: None
    r = orc.train_step(osd, b, opt, 4, 0.5, WEIGHTS)
    orc.encoder_forward = orig
    return r, store
(r32, s32), (r64, s64) = run(torch.float32), run(torch.float64)
for pi, nm in enumerate(["ctx v1", "ctx v2", "tgt v1", "tgt v2"]):
    for s in range(4):
        print(nm, s, "dfeat o32-o64 %.2e   feat %.2e" % (rel(s32[pi][s].grad, s64[pi][s].grad), rel(s32[pi][s], s64[pi][s])))
g32, g64 = r32[3], r64[3]
for k in ["context_encoder.layer4.1.conv2.weight", "context_encoder.layer4.1.bn2.weight", "context_encoder.layer4.0.conv1.weight", "context_encoder.layer3.1.conv2.weight", "context_encoder.layer3.0.conv1.weight","context_encoder.layer2.1.conv2.weight", "context_projector.3.0.weight", "inter_projector.3.0.weight", "inter_projector.0.0.weight"]:
    print(k, "%.2e" % rel(g32[k], g64[k]))
