"""Weight gradients dW = dY^T X of the block's projections at the headline training shape (M = B L = 65536 rows contracted): one library GEMM
against a batched GEMM over S row slabs (split-K by the library's batch dimension) + an fp32 sum of the S partial products."""
import json, os, sys, torch
dev, dt = "cuda", torch.bfloat16
M = 65536
torch.manual_seed(0)
def timeit(fn, n=10):
    for _ in range(2): fn()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n * 1e3
for name, N, K in (("in_proj", 2560, 640), ("out_proj", 640, 1280), ("to_q", 512, 640), ("to_out", 640, 512), ("x_proj", 72, 1280), ("dt_proj", 1280, 40)):
    dy = torch.randn(M, N, device=dev, dtype=dt); x = torch.randn(M, K, device=dev, dtype=dt)
    ref = (dy.t() @ x)
    res = dict(shape=f"{name}: dW ({N} x {K}) = dY^T ({N} x {M}) X ({M} x {K})", GF=2.0 * M * N * K / 1e9, plain_us=timeit(lambda: dy.t() @ x))
    for S in (8, 16, 32, 64):
        f = lambda: torch.bmm(dy.view(S, M // S, N).transpose(1, 2), x.view(S, M // S, K)).sum(0, dtype=torch.float32).to(dt)
        out = f()
        res[f"split{S}_us"] = timeit(f)
        res[f"split{S}_relerr_vs_plain"] = float((out.float() - ref.float()).norm() / ref.float().norm())
    print(json.dumps(res), flush=True)
