"""linear_ws_kernel (weights stationary in registers, only the tokens stream) against the 4-wave kernel and the library at the k = 640 / 512
projection shapes of the headline block: correctness (full compare with the 4-wave kernel, float64 on sampled rows), interleaved timing,
and the phase probes of the probe build (no epilogue; ring filled once = no activation stream)."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PROBE_LIB = os.path.join(ROOT, "tools", "libzigma_l4w_probes.so")
import torch  # noqa: E402
if not os.path.exists(PROBE_LIB):
    from zigma_amd import build as zbuild
    zbuild.build(verbose=False, lib=PROBE_LIB, extra_flags=("-DZIGMA_LINEAR4W_PROBES",))
os.environ["ZIGMA_AMD_LIB"] = PROBE_LIB
from zigma_amd import _lib
from zigma_amd.linear import linear
F = torch.nn.functional
dev, dt = "cuda", torch.bfloat16
M = int(os.environ.get("M", 65536))
out_path = os.path.join(ROOT, "gpurun_out", "linear_ws_probe.jsonl")
os.makedirs(os.path.dirname(out_path), exist_ok=True)
open(out_path, "w").close()
torch.manual_seed(0)


def timed(fn, n=10):
    for _ in range(2): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for name, K, N in (("in_proj", 640, 2560), ("to_q", 640, 512), ("k512_n1024", 512, 1024)):
    x = torch.randn(M, K, device=dev, dtype=dt)
    w = (torch.randn(N, K, device=dev) * K ** -0.5).to(dt)
    y4 = linear(x, w)
    k4 = _lib.last_kernel()
    yw = linear(x, w, _probe_flags=0x4000)
    kw = _lib.last_kernel()
    rows = torch.tensor([0, 1, 63, 64, 255, 256, 31337 % M, M - 1], device=dev)
    ref = x[rows].double() @ w.double().T
    err = float((yw[rows].double() - ref).norm() / ref.norm())
    diff = (yw.float() - y4.float()).abs()
    rec = dict(shape=f"{name} M={M} K={K} N={N}", kernels=[k4, kw], rel_err_vs_f64_rows=err, equal_4w=bool(torch.equal(yw, y4)),
               max_abs_diff_4w=float(diff.max()), frac_diff_4w=float((diff > 0).float().mean()), finite=bool(torch.isfinite(yw.float()).all()))
    if not rec["equal_4w"]:
        bad = (diff > 0).nonzero()
        rr, cc = bad[:, 0], bad[:, 1]
        rec["bad"] = dict(n=int(bad.shape[0]), rows_mod64=sorted(set((rr % 64).tolist()))[:70], cols_mod64=sorted(set((cc % 64).tolist()))[:8],
                          n_cols_mod64=len(set((cc % 64).tolist())), tiles_in_xcd=sorted(set(((rr // 64) % (M // 512)).tolist()))[:40],
                          first=bad[:6].tolist())
    # everything in ONE interleaved schedule (a variant timed alone runs on a cooler chip than one timed between two other kernels: 175 vs 185 us)
    PROBES = (("no_epilogue", 0x10000), ("default_policy_stores", 0x20000), ("mfma_only", 0x30000), ("frag_reads_in_front", 0x40000))
    fns = [("ws", lambda: linear(x, w, _probe_flags=0x4000)), ("4w", lambda: linear(x, w)), ("lib", lambda: F.linear(x, w))]
    fns += [(pn, (lambda fl: (lambda: linear(x, w, _probe_flags=0x4000 | fl)))(fl)) for pn, fl in PROBES]
    t = {k: [] for k, _ in fns}
    for rnd in range(5):
        for which, fn in fns:
            t[which].append(timed(fn))
    if N == 2560:                           # the default path: two half-width launches of the 4-wave kernel
        o2 = torch.empty(M, N, device=dev, dtype=dt)
        def halves():
            linear(x, w[:N // 2], out=o2[:, :N // 2]); linear(x, w[N // 2:], out=o2[:, N // 2:])
        t["4w_halves"] = [timed(halves) for _ in range(5)]
    rec["us"] = {k: sorted(v)[len(v) // 2] for k, v in t.items()}
    rec["us_min"] = {k: min(v) for k, v in t.items()}
    rec["probes_us"] = {pn: rec["us"].pop(pn) for pn, _ in PROBES}
    for pn, _ in PROBES:
        rec["us_min"].pop(pn, None)
    fl = 2.0 * M * K * N
    rec["PFLOPs"] = {k: fl / (v * 1e-6) / 1e15 for k, v in rec["us"].items()}
    print(json.dumps(rec), flush=True)
    with open(out_path, "a") as f:
        f.write(json.dumps(rec) + "\n")
