"""zigma_linear_fwd against the library (F.linear -> hipBLASLt) at the four projection shapes of the headline block
(M = 65 536 tokens, bf16): correctness vs a float64 evaluation on sampled rows, interleaved timing, TFLOP/s."""
import json, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zigma_amd import _lib
from zigma_amd.linear import linear
F = torch.nn.functional
dev, dt = "cuda", torch.bfloat16
M = int(os.environ.get("M", 65536))
torch.manual_seed(0)
res = []
for name, K, N, act in (("in_proj", 640, 2560, 1280), ("out_proj", 1280, 640, None), ("to_q", 640, 512, None), ("to_out", 512, 640, None)):
    x = torch.randn(M, K, device=dev, dtype=dt)
    w = (torch.randn(N, K, device=dev) * K ** -0.5).to(dt)
    b = (torch.randn(N, device=dev) * 0.1).to(dt) if name == "to_out" else None
    y = linear(x, w, b, act)
    kern = _lib.last_kernel()
    rows = torch.tensor([0, 1, 255, 256, 31337 % M, M - 1], device=dev)
    ref = x[rows].double() @ w.double().T + (b.double() if b is not None else 0)
    if act is not None:
        ref[:, act:] = torch.nn.functional.silu(ref[:, act:])
    err = float((y[rows].double() - ref).norm() / ref.norm())
    y3 = linear(x, w, b, act, _probe_flags=0x1000)
    err3 = float((y3.float() - y.float()).abs().max())
    lib = F.linear(x, w, b)
    if act is not None:
        lib[:, act:] = F.silu(lib[:, act:].float()).to(dt)
    err_lib = float((y.float() - lib.float()).norm() / lib.float().norm())
    t = {"own": [], "lib": [], "own_narrow3": [], "own_narrow2": []}
    for rnd in range(5):
        for which, fn in (("own", lambda: linear(x, w, b, act)), ("lib", lambda: F.linear(x, w, b)),
                          ("own_narrow3", (lambda: linear(x, w, b, act, _probe_flags=0x1000))),
                          ("own_narrow2", (lambda: linear(x, w, b, act, _probe_flags=0x1800)))):
            for _ in range(2): fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10): fn()
            e1.record(); torch.cuda.synchronize()
            t[which].append(e0.elapsed_time(e1) / 10 * 1e3)
    probes = {}
    for pname, fl_ in (("no_mfma", 0x100), ("no_loads", 0x200), ("no_stores", 0x400), ("loads_only", 0x500), ("mfma_only", 0x600)):
        fn = lambda: linear(x, w, b, act, _probe_flags=fl_)
        for _ in range(2): fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): fn()
        e1.record(); torch.cuda.synchronize()
        probes[pname] = e0.elapsed_time(e1) / 10 * 1e3
    fl = 2.0 * M * K * N
    med = {k: sorted(v)[len(v) // 2] for k, v in t.items()}
    res.append(dict(shape=f"{name} M={M} K={K} N={N}", kernel=kern, rel_err_vs_f64_rows=err, rel_err_vs_library=err_lib, maxdiff_narrow_vs_own=err3,
                    us=med, probes_us=probes, TFLOPs={k: fl / (v * 1e-6) / 1e12 for k, v in med.items()}))
    print(json.dumps(res[-1]), flush=True)
