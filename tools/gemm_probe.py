"""Time the library GEMMs of one ZigMa layer in the layouts hipBLASLt can be handed (TN = F.linear, NN = x @ W_kn)."""
import os, sys, json, torch
import torch.nn.functional as F
dev, dt = "cuda", torch.bfloat16
M = 65536
shapes = {"in_proj": (640, 2560), "out_proj": (1280, 640), "q_proj": (640, 640), "x_proj": (1280, 72), "attn_out": (640, 640)}
def timeit(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / iters * 1e3
res = {}
for name, (K, N) in shapes.items():
    x = torch.randn(M, K, device=dev, dtype=dt); W = torch.randn(N, K, device=dev, dtype=dt) * 0.02
    Wkn = W.t().contiguous(); b = torch.randn(N, device=dev, dtype=dt); out = torch.empty(M, N, device=dev, dtype=dt)
    r = {}
    r["linear_TN"] = timeit(lambda: F.linear(x, W))
    r["linear_TN_bias"] = timeit(lambda: F.linear(x, W, b))
    r["mm_NN"] = timeit(lambda: torch.mm(x, Wkn, out=out))
    r["addmm_NN"] = timeit(lambda: torch.addmm(b, x, Wkn, out=out))
    xt = x.t().contiguous()          # (K, M): the "column-major activations" variant
    r["mm_TN_xT"] = timeit(lambda: torch.mm(xt.t(), Wkn, out=out))
    for parts in (2, 4):
        xs = x.chunk(parts); os_ = out.chunk(parts)
        r[f"linear_TN_{parts}chunks"] = timeit(lambda: [torch.mm(a, W.t(), out=o) for a, o in zip(xs, os_)])
    fl = 2.0 * M * K * N
    r = {k: round(v, 1) for k, v in r.items()}
    r["PFLOPs_best"] = round(fl / (min(v for v in r.values()) * 1e-6) / 1e15, 3)
    res[name] = r
    print(name, (K, N), json.dumps(r), flush=True)
