import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch, numpy as np
from helpers import *
from oracle import msfwsi_oracle as orc
vec, man = load_golden("r18_b8_s64")
B, size = man["B"], man["size"]
model = build_product("resnet18")
sd0 = {k: v.detach().clone() for k, v in model.state_dict().items()}
batch = orc.synthetic_batch(B, size, 16, 0)
def run(dt):
    osd = {k: (v.to(dt) if v.is_floating_point() else v.clone()) for k, v in sd0.items()}
    (c1, c2), (t1, t2), idx = batch
    b = ((c1.to(dt), c2.to(dt)), (t1.to(dt), t2.to(dt)), idx)
    opt = orc.Adam(osd, [1e-3]*3); opt.step = lambda *a, **k: None
    return orc.train_step(osd, b, opt, 4, 0.5, WEIGHTS)
l32, t32, o32, g32 = run(torch.float32)
l64, t64, o64, g64 = run(torch.float64)
model = model.cuda().train()
(c1, c2), (t1, t2), idx = batch
outs = model((c1.cuda(), t1.cuda()), (c2.cuda(), t2.cuda()), idx)
loss, terms = reference_loop_loss(outs)
loss.backward(); torch.cuda.synchronize()
spread = dict(zip(man["param_keys"], vec["spread_grad"]))
print("loss", loss.item(), l32.item(), l64.item(), float(vec["loss"][0]), float(vec["loss_fp32"][0]))
rows = []
for n, p in model.named_parameters():
    rows.append((n, rel(p.grad, g32[n]), rel(p.grad, g64[n]), rel(g32[n], g64[n]), spread[n]))
rows.sort(key=lambda r: -r[2])
for r in rows[:25]: print("%-50s p-o32 %.2e  p-o64 %.2e  o32-o64 %.2e  gold %.2e" % r)
print("median p-o64", np.median([r[2] for r in rows]), "median o32-o64", np.median([r[3] for r in rows]))
fo, f32, f64 = flat_outputs(outs), flat_outputs(o32), flat_outputs(o64)
for k in fo:
    if k[0] == "context":
        print(k, "p-o32 %.2e p-o64 %.2e o32-o64 %.2e" % (rel(fo[k], f32[k]), rel(fo[k], f64[k]), rel(f32[k], f64[k])))
print("terms p", terms.cpu().numpy()[:4]); print("t32", torch.stack([t for r in t32 for t in r]).numpy()[:4]); print("t64", torch.stack([t for r in t64 for t in r]).numpy()[:4])
import subprocess
print(subprocess.run("lscpu | egrep 'Model name|^CPU\\(s\\)|Thread|Socket|Flags' | cut -c1-300; free -g | head -2", shell=True, capture_output=True, text=True).stdout)
print(torch.get_num_threads(), torch.backends.mkldnn.is_available(), torch.get_float32_matmul_precision())
print("---- vs golden (container fp64 reference)")
gn = dict(zip(man["param_keys"], vec["grad_norm"]))
for k in ["context_encoder.bn1.weight", "context_encoder.bn1.bias"]:
    gold = torch.as_tensor(vec["grad/" + k])
    pg = dict(model.named_parameters())[k].grad
    print(k, "p-gold %.2e o32-gold %.2e o64-gold %.2e" % (rel(pg, gold), rel(g32[k], gold), rel(g64[k], gold)))
for k in ["context_encoder.conv1.weight", "context_encoder.layer2.1.conv1.weight", "target_encoder.conv1.weight", "inter_projector.0.0.weight"]:
    pg = dict(model.named_parameters())[k].grad
    print(k, "norm p %.8e o32 %.8e o64 %.8e gold %.8e" % (pg.double().norm().item(), g32[k].double().norm().item(), g64[k].double().norm().item(), gn[k]))
