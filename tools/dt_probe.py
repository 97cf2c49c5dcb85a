import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zigma_amd.selective_scan_interface import dt_proj_softplus
dev, dt = "cuda", torch.bfloat16
B, L, Di, R, N = 64, 1024, 1280, 40, 16
torch.manual_seed(0)
xdbl = torch.randn(B, L, R + 2 * N, device=dev, dtype=dt); w = (torch.randn(Di, R, device=dev) * R ** -0.5).to(dt); db = torch.rand(Di, device=dev)
def timeit(fn, iters=30):
    for _ in range(5): fn()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / iters * 1e3
print("softplus on ", round(timeit(lambda: dt_proj_softplus(xdbl, R, w, db, True)), 1), "us")
print("softplus off", round(timeit(lambda: dt_proj_softplus(xdbl, R, w, db, False)), 1), "us")
print("empty alloc ", round(timeit(lambda: torch.empty(B * L, Di, device=dev, dtype=dt)), 1), "us")
