import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zigma_amd import _lib
from zigma_amd.selective_scan_interface import scan_bwd_tok
dev = "cuda"
orig = _lib.call
flag = [0]
def call(name, P, d):
    if name == "zigma_selective_scan_bwd": P.flags = flag[0]
    return orig(name, P, d)
_lib.call = call
dt = torch.bfloat16
B, L, Di, N = int(os.environ.get("B", 64)), 1024, 1280, 16
torch.manual_seed(0)
u = torch.randn(B, L, Di, device=dev, dtype=dt); delta = (0.5 * torch.rand(B, L, Di, device=dev)).to(dt)
z = torch.randn(B, L, Di, device=dev, dtype=dt); dout = torch.randn(B, L, Di, device=dev, dtype=dt)
out = torch.randn(B, L, Di, device=dev, dtype=dt)
A = (-0.5 * torch.rand(Di, N, device=dev) - 0.05); Bm = torch.randn(B, L, N, device=dev, dtype=dt); Cm = torch.randn(B, L, N, device=dev, dtype=dt)
D = torch.randn(Di, device=dev); db = torch.rand(Di, device=dev)
fn = lambda: scan_bwd_tok(u, delta, A, Bm, Cm, D, z, db, dout, out, True)
for f, name in ((0, "full"), (1, "no phase 1"), (2, "no cross-channel sums"), (4, "no reverse math"), (5, "recompute+prologue only"), (7, "7")):
    flag[0] = f
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): fn()
    e1.record(); torch.cuda.synchronize()
    print(name, round(e0.elapsed_time(e1) / 5 * 1e3, 1), "us", flush=True)
