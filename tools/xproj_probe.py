import os, sys, torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zigma_amd.selective_scan_interface import x_proj
dev, dt = "cuda", torch.bfloat16
torch.manual_seed(0)
u = (torch.randn(64, 1024, 1280, device=dev) * 0.5).to(dt); w = (torch.randn(72, 1280, device=dev) * 1280 ** -0.5).to(dt)
def timeit(fn, iters=30):
    for _ in range(5): fn()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / iters * 1e3
a, b = x_proj(u, w), F.linear(u, w)
print("rel diff vs library", ((a.float() - b.float()).norm() / b.float().norm()).item())
print("x_proj_mfma", round(timeit(lambda: x_proj(u, w)), 1), "us   library", round(timeit(lambda: F.linear(u, w)), 1), "us")
