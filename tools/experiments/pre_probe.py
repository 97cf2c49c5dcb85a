import os, sys, json, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zigma_amd import _lib
import zigma_amd.selective_scan_interface as ssi
dev, dt = "cuda", torch.bfloat16
B, L, Di, N, R = 64, 1024, 1280, 16, 40
torch.manual_seed(0)
xz = torch.randn(B, L, 2 * Di, device=dev, dtype=dt)
cw4, cb4 = torch.randn(Di, 4, device=dev, dtype=dt) * 0.5, torch.randn(Di, device=dev, dtype=dt) * 0.2
wx = (torch.randn(R + 2 * N, Di, device=dev) * Di ** -0.5).to(dt); wd = (torch.randn(Di, R, device=dev) * R ** -0.5).to(dt)
db = torch.rand(Di, device=dev); perm = torch.randperm(L, device=dev).to(torch.int32)
orig = _lib.call
flag = [0]
def call(name, P, d):
    if name == "zigma_mamba_pre_fwd": P.flags = flag[0]
    return orig(name, P, d)
_lib.call = call
def timeit(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / iters * 1e3
for f, name in ((0, "full"), (1, "no x_proj mfma"), (2, "no dt phase"), (3, "conv only"), (7, "conv only, no u store"), (4, "no u store")):
    flag[0] = f
    print(name, round(timeit(lambda: ssi.mamba_pre(xz[:, :, :Di], cw4, cb4, wx, wd, db, N, perm=perm)), 1), "us", flush=True)
