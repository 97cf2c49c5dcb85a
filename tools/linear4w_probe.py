"""linear4w_kernel (one wave per SIMD, generated main loop) against the 8-wave kernel and the library at the wide projection shapes of
the headline block: correctness (vs float64 on sampled rows, bit-identity with the 8-wave kernel), interleaved timing, PFLOP/s."""
import json, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zigma_amd import _lib
from zigma_amd.linear import linear
F = torch.nn.functional
dev, dt = "cuda", torch.bfloat16
M = int(os.environ.get("M", 65536))
torch.manual_seed(0)
for name, K, N in (("in_proj", 640, 2560), ("to_q", 640, 512), ("square", 1024, 1024), ("k1280", 1280, 1280)):
    x = torch.randn(M, K, device=dev, dtype=dt)
    w = (torch.randn(N, K, device=dev) * K ** -0.5).to(dt)
    y = linear(x, w)
    kern = _lib.last_kernel()
    y8 = linear(x, w, _probe_flags=0x2000)
    rows = torch.tensor([0, 1, 255, 256, 31337 % M, M - 1], device=dev)
    ref = x[rows].double() @ w.double().T
    err = float((y[rows].double() - ref).norm() / ref.norm())
    t = {"4w": [], "8w": [], "lib": []}
    for rnd in range(5):
        for which, fn in (("4w", lambda: linear(x, w)), ("8w", lambda: linear(x, w, _probe_flags=0x2000)), ("lib", lambda: F.linear(x, w))):
            for _ in range(2): fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10): fn()
            e1.record(); torch.cuda.synchronize()
            t[which].append(e0.elapsed_time(e1) / 10 * 1e3)
    fl = 2.0 * M * K * N
    med = {k: sorted(v)[len(v) // 2] for k, v in t.items()}
    print(json.dumps(dict(shape=f"{name} M={M} K={K} N={N}", kernel=kern, rel_err_vs_f64_rows=err, equal_8w=bool(torch.equal(y, y8)),
                          us=med, us_min={k: min(v) for k, v in t.items()}, PFLOPs={k: fl / (v * 1e-6) / 1e15 for k, v in med.items()},
                          GBps_4w=(M * K + N * K + M * N) * 2 / (med["4w"] * 1e-6) / 1e9)), flush=True)
