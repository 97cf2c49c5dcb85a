"""Round 6: what does N = 640 = 2.5 tiles of 256 cost the 4-wave tiled kernel?  The same kernel at N = 512 (2 full tiles), 640 (2 + a
128-wide one) and 768 (3 full tiles), K = 1280 (out_proj) and 512 (to_out), 65 536 tokens, plain and with the gated-add (+ bias) epilogue: if the
half tile were as dear as a full one, N = 640 would take the time of N = 768 and a 320-wide tile would save 17 %; if time follows the columns, it saves nothing."""
import json, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zigma_amd import _lib
from zigma_amd.linear import linear
dev, dt = "cuda", torch.bfloat16
Bsz, L = 64, 1024
g = torch.Generator().manual_seed(0)
res = []
cases = []
for K in (1280, 512):
    for N in (512, 640, 768):
        x = torch.randn(Bsz, L, K, generator=g).to(dev, dt)
        w = (torch.randn(N, K, generator=g) * K ** -0.5).to(dev, dt)
        b = (torch.randn(N, generator=g) * 0.1).to(dev, dt)
        r = torch.randn(Bsz, L, N, generator=g).to(dev, dt)
        gt = torch.randn(Bsz, N, generator=g).to(dev, dt)
        cases.append((K, N, x, w, b, r, gt))
def t_of(fn, n=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for rnd in range(3):
    for K, N, x, w, b, r, gt in cases:
        row = dict(round=rnd, K=K, N=N)
        row["plain_us"] = t_of(lambda: linear(x, w)); row["kernel_plain"] = _lib.last_kernel()
        row["gated_us"] = t_of(lambda: linear(x, w, residual=r, gate=gt)); row["kernel_gated"] = _lib.last_kernel()
        row["bias_gated_us"] = t_of(lambda: linear(x, w, b, residual=r, gate=gt))
        row["lib_us"] = t_of(lambda: torch.nn.functional.linear(x, w))
        fl = 2 * Bsz * L * K * N
        row["plain_PFs"], row["gated_PFs"] = fl / row["plain_us"] / 1e9, fl / row["gated_us"] / 1e9
        res.append(row)
        print(json.dumps(row), flush=True)
