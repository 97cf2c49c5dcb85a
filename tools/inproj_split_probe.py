"""in_proj (M=65536, K=640, N=2560) on linear4w_kernel as ONE launch vs column slabs of the weight in consecutive launches
(each XCD's L2 then holds a slab of W + its activation panels instead of thrashing on the whole W), against the library."""
import json, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zigma_amd.linear import linear
F = torch.nn.functional
dev, dt = "cuda", torch.bfloat16
M, K, N = int(os.environ.get("M", 65536)), int(os.environ.get("K", 640)), int(os.environ.get("N", 2560))
torch.manual_seed(0)
x = torch.randn(M, K, device=dev, dtype=dt); w = (torch.randn(N, K, device=dev) * K ** -0.5).to(dt)
out = torch.empty(M, N, device=dev, dtype=dt)
def split(parts):
    step = N // parts
    for i in range(parts):
        linear(x, w[i * step:(i + 1) * step], out=out[:, i * step:(i + 1) * step])
PARTS = [int(p_) for p_ in os.environ.get("PARTS", "1,2,5,10").split(",")]
variants = {**{f"4w_{p_}": (lambda p_=p_: split(p_)) for p_ in PARTS}, "lib": lambda: F.linear(x, w)}
ref = F.linear(x, w)
ok = {}
for k_, fn in variants.items():
    if k_ != "lib":
        out.zero_(); fn(); ok[k_] = bool(torch.equal(out, ref)) or float((out.float() - ref.float()).abs().max())
t = {k_: [] for k_ in variants}
for rnd in range(5):
    for k_, fn in variants.items():
        for _ in range(2): fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): fn()
        e1.record(); torch.cuda.synchronize()
        t[k_].append(e0.elapsed_time(e1) / 10 * 1e3)
print(json.dumps(dict(shape=f"M={M} K={K} N={N}", us_median={k_: sorted(v)[2] for k_, v in t.items()}, us_min={k_: min(v) for k_, v in t.items()}, same_as_lib=ok)))
