"""Symmetric A/B of two variants of the weight-stationary kernel (probe build): A, B, A, B ... with identical predecessors, 10 launches per sample.
usage: python tools/linear_ws_ab.py <probe flag of B in hex, e.g. 0x40000> [K N]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["ZIGMA_AMD_LIB"] = os.path.join(ROOT, "tools", "libzigma_l4w_probes.so")
import torch
from zigma_amd.linear import linear
flag = int(sys.argv[1], 16) if len(sys.argv) > 1 else 0x40000
K, N = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (640, 2560)
M = 65536
x = torch.randn(M, K, device="cuda", dtype=torch.bfloat16)
w = (torch.randn(N, K, device="cuda") * K ** -0.5).bfloat16()
fa = lambda: linear(x, w, _probe_flags=0x4000)
fb = lambda: linear(x, w, _probe_flags=0x4000 | flag)
def timed(fn, n=10):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
ya, yb = fa(), fb()
torch.cuda.synchronize()
equal = bool(torch.equal(ya, yb))
for _ in range(3): fa(); fb()
ta, tb = [], []
for _ in range(12):
    ta.append(timed(fa)); tb.append(timed(fb))
med = lambda v: sorted(v)[len(v) // 2]
print(json.dumps(dict(shape=f"M={M} K={K} N={N}", flag=hex(flag), equal=equal, finite=bool(torch.isfinite(yb.float()).all()), frac_diff=float((ya != yb).float().mean()), default_us=med(ta), variant_us=med(tb), default_min=min(ta), variant_min=min(tb))))
