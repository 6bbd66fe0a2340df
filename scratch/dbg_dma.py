import os, sys, torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from msf_wsi_amd import kernels as kn, _lib
lib = _lib.load()
torch.manual_seed(0)
for dt in (torch.bfloat16, torch.float32):
    for (N, H, C, K, R, st) in ((2, 14, 64, 64, 1, 1), (2, 14, 64, 64, 3, 1), (3, 13, 32, 128, 3, 2)):
        x = torch.randn(N, H, H, C).to(dt)
        w = (torch.randn(K, R, R, C) * 0.1).to(dt)
        d = kn.conv_desc(dt, N, H, H, C, K, R, R, st, R // 2)
        ref = F.conv2d(x.float().permute(0, 3, 1, 2), w.float().permute(0, 3, 1, 2), stride=st, padding=R // 2).permute(0, 2, 3, 1)
        for fast in (0, 1):
            lib.msfwsi_set_tuning(1, fast)
            y = torch.zeros(N, d.P, d.Q, K, dtype=dt, device="cuda")
            kn.conv_fwd(d, x.cuda(), w.cuda(), y)
            torch.cuda.synchronize()
            e = (y.float().cpu() - ref).norm() / ref.norm()
            print(dt, (N, H, C, K, R, st), "fast", fast, "rel", float(e), "y[0,0,0,:4]", y[0, 0, 0, :4].float().cpu().tolist(), "ref", ref[0, 0, 0, :4].tolist(), flush=True)
print("---- wgrad ----")
for dt in (torch.bfloat16, torch.float32):
    for (N, H, C, K, R, st) in ((2, 14, 64, 64, 1, 1), (2, 14, 64, 64, 3, 1), (5, 7, 128, 96, 3, 1), (3, 2, 64, 64, 3, 1), (300, 9, 32, 128, 3, 1)):
        x = torch.randn(N, H, H, C).to(dt)
        d = kn.conv_desc(dt, N, H, H, C, K, R, R, st, R // 2)
        dy = torch.randn(N, d.P, d.Q, K).to(dt)
        ref = torch.nn.grad.conv2d_weight(x.float().permute(0, 3, 1, 2), (K, C, R, R), dy.float().permute(0, 3, 1, 2), stride=st, padding=R // 2).permute(0, 2, 3, 1)
        for fast in (0, 1):
            lib.msfwsi_set_tuning(2, fast)
            dw = torch.zeros(K, R, R, C, device="cuda")
            kn.conv_wgrad(d, x.cuda(), dy.cuda(), dw)
            torch.cuda.synchronize()
            e = (dw.cpu() - ref).norm() / ref.norm()
            print(dt, (N, H, C, K, R, st), "lin", fast, "rel", float(e), flush=True)
