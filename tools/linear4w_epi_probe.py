"""out_proj / to_out with the block's gated add: 4-wave kernel against the 8-wave kernel and against the library GEMM + torch.addcmul"""
import json, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zigma_amd import _lib
import zigma_amd.linear as zl
from zigma_amd.linear import linear
# (linear() carries no policy: routing lives in zigma_amd/routing.py)
F = torch.nn.functional
dev, dt = "cuda", torch.bfloat16
B, L = int(os.environ.get("B", 64)), 1024
torch.manual_seed(0)
for name, K, N, bias, res in (("out_proj+add", 1280, 640, False, True), ("to_out+bias+add", 512, 640, True, True), ("out_proj", 1280, 640, False, False)):
    x = torch.randn(B, L, K, device=dev, dtype=dt)
    w = (torch.randn(N, K, device=dev) * K ** -0.5).to(dt)
    b = (torch.randn(N, device=dev) * 0.1).to(dt) if bias else None
    r = torch.randn(B, L, N, device=dev, dtype=dt) if res else None
    g = torch.randn(B, N, device=dev, dtype=dt) if res else None
    y = linear(x, w, b, residual=r, gate=g); kern = _lib.last_kernel()
    fns = {"4w": lambda: linear(x, w, b, residual=r, gate=g), "8w": lambda: linear(x, w, b, residual=r, gate=g, _probe_flags=0x2000),
           "lib": (lambda: torch.addcmul(r, g.unsqueeze(1), F.linear(x, w, b))) if res else (lambda: F.linear(x, w, b)),
           "lib_gemm_only": lambda: F.linear(x, w, b)}
    t = {k: [] for k in fns}
    for rnd in range(5):
        for k, fn in fns.items():
            for _ in range(2): fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10): fn()
            e1.record(); torch.cuda.synchronize()
            t[k].append(e0.elapsed_time(e1) / 10 * 1e3)
    ref = fns["lib"]()
    print(json.dumps(dict(shape=f"{name} B={B} K={K} N={N}", kernel=kern, rel_vs_lib=float((y.float() - ref.float()).norm() / ref.float().norm()),
                          us={k: sorted(v)[2] for k, v in t.items()})), flush=True)
