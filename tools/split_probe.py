import os, sys, json, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zigma_amd.selective_scan_interface import scan_raw
dev, dt = "cuda", torch.bfloat16
B, L, Di, N, R = 4, 16384, 1280, 16, 40
torch.manual_seed(0)
xz = torch.randn(B, L, 2 * Di, device=dev, dtype=dt); u = torch.randn(B, L, Di, device=dev, dtype=dt)
delta = torch.rand(B, L, Di, device=dev).to(dt); xdbl = torch.randn(B, L, R + 2 * N, device=dev, dtype=dt)
A = -torch.exp(torch.log(torch.arange(1, N + 1, device=dev).float())).repeat(Di, 1).contiguous()
D = torch.randn(Di, device=dev); perm = torch.randperm(L, device=dev).to(torch.int32)
y = torch.empty(B, L, Di, device=dev, dtype=dt)
Bv = xdbl[:, :, R:R + N].transpose(1, 2).unsqueeze(1); Cv = xdbl[:, :, R + N:].transpose(1, 2).unsqueeze(1)
xc = torch.empty(B, Di, (L + 2047) // 2048, 2 * N, device=dev)
def run(x, p):
    scan_raw(u.transpose(1, 2), delta.transpose(1, 2), A, Bv, Cv, D, xz[:, :, Di:].transpose(1, 2), None, False,
             out_z=y.transpose(1, 2), z_row_index=p, out_row_index=p, want_out=False, x=x)
def timeit(fn, iters=5):
    fn(); torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / iters * 1e3
for name, x, p in (("single,perm", None, perm), ("split,perm", xc, perm), ("single,noperm", None, None), ("split,noperm", xc, None)):
    print(name, round(timeit(lambda: run(x, p)), 1), "us")
y1 = y.clone(); run(None, perm); y2 = y.clone(); run(xc, perm)
print("split vs single rel diff", ((y.float() - y2.float()).norm() / y2.float().norm()).item())
