import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zigma_amd.layernorm import block_norm
dev, dt = "cuda", torch.bfloat16
B, L, E = 64, 1024, 640
torch.manual_seed(0)
x = torch.randn(B, L, E, device=dev, dtype=dt); br = torch.randn(B, L, E, device=dev, dtype=dt)
res = torch.randn(B, L, E, device=dev); w = torch.ones(E, device=dev, dtype=dt)
mod = torch.randn(B, 6 * E, device=dev, dtype=dt)
def timeit(fn, iters=30):
    for _ in range(5): fn()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / iters * 1e3
n1 = lambda: block_norm(x, w, None, res, 1e-5, True, residual_in_fp32=True, branch=br, gate=mod[:, 0:E], shift=mod[:, E:2*E], scale=mod[:, 2*E:3*E])
n2 = lambda: block_norm(x, None, None, None, 1e-6, False, residual_in_fp32=False, branch=br, gate=mod[:, 0:E], shift=mod[:, E:2*E], scale=mod[:, 2*E:3*E], want_x=True, want_y=False, want_res_out=False)
print("norm1 (rms, f32 residual)", round(timeit(n1), 1), "us   norm2 (ln, bf16 only)", round(timeit(n2), 1), "us")
# the forms the block uses since the gated adds moved into the projection epilogues: no branch on the way in
n1b = lambda: block_norm(x, w, None, res, 1e-5, True, residual_in_fp32=True, shift=mod[:, E:2*E], scale=mod[:, 2*E:3*E])
n2b = lambda: block_norm(x, None, None, None, 1e-6, False, residual_in_fp32=False, shift=mod[:, E:2*E], scale=mod[:, 2*E:3*E], want_x=False, want_y=False, want_res_out=False)
print("norm1 no branch", round(timeit(n1b), 1), "us   norm2 no branch", round(timeit(n2b), 1), "us")
