import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from msf_wsi_amd import kernels as kn
from msf_wsi_amd import _lib as _L
_L.load().msfwsi_set_tuning(4, int(os.environ.get("SMALL", "640")))
dt = torch.bfloat16
def timeit(fn, rep=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(rep): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / rep
for (M, C, K) in ((256, 18432, 18432), (512, 18432, 18432), (256, 9216, 9216), (512, 9216, 9216), (256, 4608, 4608), (512, 4608, 4608), (4096, 2048, 2048), (8192, 2048, 2048), (256, 18432, 4608), (512, 18432, 4608)):
    d = kn.conv_desc(dt, M, 1, 1, C, K, 1, 1, 1, 0)
    x = torch.randn(M, 1, 1, C, device="cuda").to(dt); w = (torch.randn(K, 1, 1, C, device="cuda") * 0.01).to(dt)
    y = torch.empty(M, 1, 1, K, dtype=dt, device="cuda"); dy = torch.randn_like(y); dx = torch.empty_like(x)
    dw = torch.zeros(K, 1, 1, C, device="cuda")
    sc, sh = torch.rand(C, device="cuda") + 0.5, torch.randn(C, device="cuda") * 0.1
    st = kn.new_stats(K)
    fl = 2.0 * M * C * K; wb = 2.0 * C * K
    tf = timeit(lambda: kn.conv_fwd(d, x, w, y, stats=st))
    tfp = timeit(lambda: kn.conv_fwd(d, x, w, y, pro=(sc, sh), stats=st))
    td = timeit(lambda: kn.conv_dgrad(d, dy, w, dx))
    tw = timeit(lambda: kn.conv_wgrad(d, x, dy, dw))
    twp = timeit(lambda: kn.conv_wgrad(d, x, dy, dw, pro=(sc, sh)))
    print(f"M={M} {C}->{K}: fwd {tf:.3f} ms ({fl/tf/1e9:.0f} TF, w {wb/tf/1e6:.0f} GB/s) fwd+pro {tfp:.3f} | dgrad {td:.3f} ({fl/td/1e9:.0f} TF) | wgrad {tw:.3f} wgrad+pro {twp:.3f} ({fl/twp/1e9:.0f} TF, dw {2*wb/twp/1e6:.0f} GB/s)", flush=True)
