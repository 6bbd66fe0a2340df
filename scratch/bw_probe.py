import torch
def timeit(fn, rep=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(rep): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / rep
n = 1 << 30  # 2 GiB bf16
x = torch.empty(n, dtype=torch.bfloat16, device="cuda").normal_()
o = torch.empty_like(x)
B = n * 2 / 1e9
t = timeit(lambda: o.zero_()); print(f"fill (pure write)  {t:.3f} ms {B/t*1e3:.0f} GB/s")
t = timeit(lambda: o.copy_(x)); print(f"copy (1R:1W)       {t:.3f} ms {2*B/t*1e3:.0f} GB/s")
t = timeit(lambda: x.sum()); print(f"sum  (pure read)   {t:.3f} ms {B/t*1e3:.0f} GB/s")
q = x[: n // 4]
t = timeit(lambda: o.view(4, -1).copy_(q.view(1, -1).expand(4, -1))); print(f"bcast (1R:4W)      {t:.3f} ms {1.25*B/t*1e3:.0f} GB/s")
t = timeit(lambda: torch.add(x, x, out=o)); print(f"add (1R... x twice) {t:.3f} ms {2*B/t*1e3:.0f} GB/s")
y = torch.empty_like(x).normal_()
t = timeit(lambda: torch.add(x, y, out=o)); print(f"add (2R:1W)        {t:.3f} ms {3*B/t*1e3:.0f} GB/s")
