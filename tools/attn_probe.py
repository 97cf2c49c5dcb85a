import os, sys, torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zigma_amd.attention import cross_attn
dev, dt = "cuda", torch.bfloat16
B, L, H, NC = 64, 1024, 8, 77
torch.manual_seed(0)
q = torch.randn(B, L, H * 64, device=dev, dtype=dt); kv = torch.randn(B, NC, 18, 2, H * 64, device=dev, dtype=dt)
k, v = kv[:, :, 3, 0], kv[:, :, 3, 1]
def timeit(fn, iters=30):
    for _ in range(5): fn()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / iters * 1e3
def sdpa():
    qq = q.view(B, L, H, 64).transpose(1, 2); kk = k.reshape(B, NC, H, 64).transpose(1, 2); vv = v.reshape(B, NC, H, 64).transpose(1, 2)
    return F.scaled_dot_product_attention(qq, kk, vv).transpose(1, 2).reshape(B, L, -1)
t1 = timeit(lambda: cross_attn(q, k, v, H)); t2 = timeit(sdpa)
by = 2 * B * L * H * 64 * 2
print(f"cross_attn_mfma {t1:.1f} us ({by / t1 / 1e6:.2f} TB/s of Q+O traffic)   torch SDPA {t2:.1f} us")
# backward: own kernel vs the GEMM + ATen composition of the same formulas vs autograd through torch's fused SDPA
from zigma_amd.attention import cross_attn_bwd, cross_attn_bwd_math
do = torch.randn_like(q)
t3 = timeit(lambda: cross_attn_bwd(q, k, v, do, H)); t4 = timeit(lambda: cross_attn_bwd_math(q, k, v, do, H, 0.125), iters=5)
qq, kk, vv = (t.clone().requires_grad_(True) for t in (q, k, v))
def sdpa_bwd():
    o = F.scaled_dot_product_attention(qq.view(B, L, H, 64).transpose(1, 2), kk.reshape(B, NC, H, 64).transpose(1, 2),
                                       vv.reshape(B, NC, H, 64).transpose(1, 2)).transpose(1, 2).reshape(B, L, -1)
    torch.autograd.grad(o, (qq, kk, vv), do)
t5 = timeit(sdpa_bwd, iters=10)
print(f"cross_attn_bwd_mfma {t3:.1f} us ({3 * by / 2 / t3 / 1e6:.2f} TB/s of Q+dO+dQ traffic)   GEMM+ATen composition {t4:.1f} us   "
      f"torch SDPA fwd+bwd {t5:.1f} us")
