"""achieved HBM bandwidth of the streaming kernels (diagnostic)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from msf_wsi_amd import kernels as kn
dt = torch.bfloat16
def timeit(fn, rep=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(rep): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / rep
for (M, C) in ((4096 * 3136, 64), (4096 * 3136, 256), (4096 * 196, 1024), (1024 * 49, 2048)):
    x = torch.randn(M, C, device="cuda").to(dt); y = torch.randn(M, C, device="cuda").to(dt); o = torch.empty_like(x)
    sc, sh = torch.rand(C, device="cuda") + 0.5, torch.randn(C, device="cuda") * 0.1
    k3 = torch.randn(C, device="cuda")
    B = x.numel() * 2 / 1e9
    t = timeit(lambda: kn.bn_act(x, sc, sh, o, relu=True)); print(f"M={M} C={C} bn_act        {t:7.3f} ms {2*B/t*1e3:7.1f} GB/s")
    t = timeit(lambda: kn.bn_act(x, sc, sh, o, ident=y, relu=True)); print(f"M={M} C={C} bn_act+ident  {t:7.3f} ms {3*B/t*1e3:7.1f} GB/s")
    sa = torch.zeros(C, dtype=torch.float64, device="cuda")
    t = timeit(lambda: kn.bn_act_sum(x, sc, sh, o, sa)); print(f"M={M} C={C} bn_act_sum    {t:7.3f} ms {2*B/t*1e3:7.1f} GB/s")
    t = timeit(lambda: kn.bn_bwd_apply(x, y, sc, sh, k3, o)); print(f"M={M} C={C} bn_bwd_apply  {t:7.3f} ms {3*B/t*1e3:7.1f} GB/s")
    s3 = kn.new_stats(C, 3, "cuda")
    t = timeit(lambda: kn.block_end_bwd(x, y, None, 1.0, None, None, o, s3, 49)); print(f"M={M} C={C} block_end_bwd {t:7.3f} ms {3*B/t*1e3:7.1f} GB/s")
    t = timeit(lambda: kn.colsum(x, sa)); print(f"M={M} C={C} colsum        {t:7.3f} ms {1*B/t*1e3:7.1f} GB/s")
    t = timeit(lambda: o.copy_(x)); print(f"M={M} C={C} torch copy    {t:7.3f} ms {2*B/t*1e3:7.1f} GB/s", flush=True)
    del x, y, o
