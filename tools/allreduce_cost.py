"""GPU box: what one tiny all-reduce of the SyncBatchNorm exchange costs on a single-rank RCCL group -- host enqueue time per
call (the launching thread feeds three streams) and GPU time per call.   python tools/allreduce_cost.py"""
import os
import time

import torch
import torch.distributed as dist

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
x = torch.zeros(512, dtype=torch.float64, device="cuda")
for _ in range(20):
    dist.all_reduce(x)
torch.cuda.synchronize()
n = 1000
t0 = time.perf_counter()
for _ in range(n):
    dist.all_reduce(x)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"all_reduce of 4 KB, one rank: host enqueue {1e6 * (t1 - t0) / n:.1f} us/call, enqueue + drain {1e6 * (t2 - t0) / n:.1f} us/call")
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(n):
    dist.all_reduce(x)
e1.record()
torch.cuda.synchronize()
print(f"GPU stream time {1e3 * e0.elapsed_time(e1) / n:.1f} us/call")
y = torch.zeros(512, dtype=torch.float64, device="cuda")
t0 = time.perf_counter()
for _ in range(n):
    y.add_(1.0)
t1 = time.perf_counter()
torch.cuda.synchronize()
print(f"for scale: a tiny eager kernel launch {1e6 * (t1 - t0) / n:.1f} us/call on the host")
dist.destroy_process_group()
