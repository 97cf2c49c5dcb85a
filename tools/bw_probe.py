import torch
dev = "cuda"
n = 64 * 1024 * 1280
a = torch.empty(n, device=dev, dtype=torch.bfloat16); b = torch.randn(n, device=dev).to(torch.bfloat16)
def timeit(fn, iters=30):
    for _ in range(5): fn()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / iters * 1e3
t = timeit(lambda: a.fill_(1.0)); print(f"fill  168 MB write          {t:.1f} us  {n*2/t/1e6:.2f} TB/s")
t = timeit(lambda: a.copy_(b));   print(f"copy  168 MB read + write   {t:.1f} us  {2*n*2/t/1e6:.2f} TB/s")
t = timeit(lambda: b.sum());      print(f"sum   168 MB read           {t:.1f} us  {n*2/t/1e6:.2f} TB/s")
big = torch.empty(8 * n, device=dev, dtype=torch.bfloat16); big2 = torch.empty_like(big)
t = timeit(lambda: big.copy_(big2)); print(f"copy  1.34 GB read + write  {t:.1f} us  {2*8*n*2/t/1e6:.2f} TB/s")
t = timeit(lambda: big.fill_(1.0)); print(f"fill  1.34 GB write         {t:.1f} us  {8*n*2/t/1e6:.2f} TB/s")
