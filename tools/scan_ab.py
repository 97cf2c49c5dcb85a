"""A/B of the token-major scan kernels at the headline shape as the model calls it (B=64, L=1024, Di=1280, N=16, bf16, delta
already softplus'ed by the dt_proj kernel, gate only, zigzag row tables): first-generation scan_tok_kernel (probe flag ZIGMA_SCAN_PROBE_V1)
against scan_tok2_kernel, interleaved rounds in ONE process (cdna_hip_programming.md §5.4 rule 24), plus the
z-preactivated variant (SiLU moved out of the kernel).  Prints one JSON line."""
import json, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PROBES = os.environ.get("PROBES") == "1"        # the no-barrier / no-hand-over timing probes live in a probe build (-DZIGMA_SCAN_PROBES)
if PROBES:
    PROBE_LIB = os.path.join(ROOT, "tools", "libzigma_scan_probes.so")
    if not os.path.exists(PROBE_LIB):           # (build it in the container before the GPU call)
        from zigma_amd import build as zbuild
        zbuild.build(verbose=False, lib=PROBE_LIB, extra_flags=("-DZIGMA_SCAN_PROBES",))
    os.environ["ZIGMA_AMD_LIB"] = PROBE_LIB
from zigma_amd import _lib
from zigma_amd.selective_scan_interface import scan_raw
dev, dt = "cuda", torch.bfloat16
B, L, Di, N, R = int(os.environ.get("B", 64)), int(os.environ.get("L", 1024)), 1280, 16, 40
torch.manual_seed(0)
xz = torch.randn(B, L, 2 * Di, device=dev, dtype=dt); u = torch.randn(B, L, Di, device=dev, dtype=dt)
delta = (0.5 * torch.rand(B, L, Di, device=dev)).to(dt); xdbl = torch.randn(B, L, R + 2 * N, device=dev, dtype=dt)
A = -torch.exp(torch.log(torch.arange(1, N + 1, device=dev).float()) + 0.1 * torch.randn(Di, N, device=dev)).contiguous()
D = torch.randn(Di, device=dev); perm = torch.randperm(L, device=dev).to(torch.int32)
Bv = xdbl[:, :, R:R + N].transpose(1, 2).unsqueeze(1); Cv = xdbl[:, :, R + N:].transpose(1, 2).unsqueeze(1)
outs = {}


db = torch.rand(Di, device=dev)


def run_sp(name):
    """softplus(delta + bias) evaluated INSIDE the kernel's prologue (what a caller without the dt_proj kernel gets)"""
    y = outs.setdefault(name, torch.empty(B, L, Di, device=dev, dtype=dt))
    scan_raw(u.transpose(1, 2), delta.transpose(1, 2), A, Bv, Cv, D, xz[:, :, Di:].transpose(1, 2), db, True,
             out_z=y.transpose(1, 2), z_row_index=perm, out_row_index=perm, want_out=False)
    return _lib.last_kernel()


def run(name, env, zact=False):
    fl = _lib.SCAN_PROBE_V1 if env == "v1" else (int(env[4:]) << _lib.SCAN_PROBE_PRIO_SHIFT) if env and env.startswith("prio") else \
        int(env[3:], 16) if env and env.startswith("raw") else 0
    y = outs.setdefault(name, torch.empty(B, L, Di, device=dev, dtype=dt))
    scan_raw(u.transpose(1, 2), delta.transpose(1, 2), A, Bv, Cv, D, xz[:, :, Di:].transpose(1, 2), None, False,
             out_z=y.transpose(1, 2), z_row_index=perm, out_row_index=perm, want_out=False, z_preactivated=zact, _probe_flags=fl)
    return _lib.last_kernel()


variants = [("v1", "v1", False), ("v2", None, False), ("v2_zact", None, True), ("v2_softplus_inside", "SP", False)] + \
           [("v2_no_prio_rotation", "prio1", False)] + \
           ([("probe_no_barriers", "raw2000", False), ("probe_no_y_handover", "raw4000", False), ("probe_neither", "raw6000", False)] if PROBES else [])
_run = run
run = lambda n, e, z=False: run_sp(n) if e == "SP" else _run(n, e, z)
names = {n: run(n, e, z) for n, e, z in variants}
torch.cuda.synchronize()
times = {n: [] for n, _, _ in variants}
for rnd in range(6):
    for n, e, z in variants:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            run(n, e, z)
        e1.record(); torch.cuda.synchronize()
        times[n].append(e0.elapsed_time(e1) / 10 * 1e3)
by = B * L * (4 * 2 * Di + 2 * 2 * N) + 4 * Di * (N + 2)
res = dict(shape=f"B={B} L={L} Di={Di} N={N} bf16", kernels=names,
           us_median={n: sorted(v)[len(v) // 2] for n, v in times.items()}, us_min={n: min(v) for n, v in times.items()},
           hbm_frac_of_8TBps={n: by / (sorted(v)[len(v) // 2] * 1e-6) / 8e12 for n, v in times.items()},
           max_abs_diff_v2_vs_v1=float((outs["v2"].float() - outs["v1"].float()).abs().max()),
           bit_identical=bool(torch.equal(outs["v2"], outs["v1"])),
           prio_variants_identical=bool(torch.equal(outs["v2"], outs["v2_no_prio_rotation"])))
print(json.dumps(res))
