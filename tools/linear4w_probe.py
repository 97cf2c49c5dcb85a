"""linear4w_kernel (one wave per SIMD, generated main loop) against the 8-wave kernel and the library at the wide projection shapes of
the headline block: correctness (vs float64 on sampled rows, bit-identity with the 8-wave kernel), interleaved timing, PFLOP/s."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
# a library with the timing-probe variants of the 4-wave kernel (not in the shipped one)
PROBE_LIB = os.path.join(ROOT, "tools", "libzigma_l4w_probes.so")
import torch  # noqa: E402  (first: the library binds to torch's HIP runtime)
if not os.path.exists(PROBE_LIB):      # (built in the container before the GPU call: python -c "from zigma_amd import build as b; b.build(lib=..., extra_flags=...)")
    from zigma_amd import build as zbuild
    zbuild.build(verbose=False, lib=PROBE_LIB, extra_flags=("-DZIGMA_LINEAR4W_PROBES",))
os.environ["ZIGMA_AMD_LIB"] = PROBE_LIB
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
    probes = {}
    for pname, fl_ in (("no_mfma", 0x10000), ("loads_only", 0x20000), ("no_glds", 0x30000), ("no_store", 0x40000), ("mfma_reads_only", 0x50000),
                       ("no_glds_lax_waits", 0x60000), ("mfma_reads_no_epilogue", 0x70000)):
        fn = lambda: linear(x, w, _probe_flags=fl_)
        for _ in range(2): fn()
        ts = []
        for rnd in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10): fn()
            e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / 10 * 1e3)
        probes[pname] = sorted(ts)[1]
    fl = 2.0 * M * K * N
    med = {k: sorted(v)[len(v) // 2] for k, v in t.items()}
    print(json.dumps(dict(shape=f"{name} M={M} K={K} N={N}", kernel=kern, rel_err_vs_f64_rows=err, equal_8w=bool(torch.equal(y, y8)),
                          us=med, probes_us=probes, us_min={k: min(v) for k, v in t.items()}, PFLOPs={k: fl / (v * 1e-6) / 1e15 for k, v in med.items()},
                          GBps_4w=(M * K + N * K + M * N) * 2 / (med["4w"] * 1e-6) / 1e9)), flush=True)
