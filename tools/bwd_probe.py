"""Stand-alone timing of the scan backward kernel at the headline shape (with the backward's own forward phase, and with
checkpoints handed over by the forward kernel)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PROBE = int(os.environ.get("PROBE", 0))      # phase-skipping timing probe: a probe build with -DZIGMA_SCANBWD_PROBE=<mask> (results wrong)
if PROBE:
    PROBE_LIB = os.path.join(ROOT, "tools", f"libzigma_scanbwd_probe{PROBE}.so")
    if not os.path.exists(PROBE_LIB):           # (build them in the container before the GPU call)
        from zigma_amd import build as zbuild
        zbuild.build(verbose=False, lib=PROBE_LIB, extra_flags=(f"-DZIGMA_SCANBWD_PROBE={PROBE}",))
    os.environ["ZIGMA_AMD_LIB"] = PROBE_LIB
from zigma_amd.selective_scan_interface import scan_bwd_tok, scan_raw
dev, dt = "cuda", torch.bfloat16
B, L, Di, N = int(os.environ.get("B", 64)), 1024, 1280, 16
torch.manual_seed(0)
u = torch.randn(B, L, Di, device=dev, dtype=dt); delta = (0.5 * torch.rand(B, L, Di, device=dev)).to(dt)
z = torch.randn(B, L, Di, device=dev, dtype=dt); dout = torch.randn(B, L, Di, device=dev, dtype=dt)
A = (-0.5 * torch.rand(Di, N, device=dev) - 0.05); Bm = torch.randn(B, L, N, device=dev, dtype=dt); Cm = torch.randn(B, L, N, device=dev, dtype=dt)
D = torch.randn(Di, device=dev); db = torch.rand(Di, device=dev)
out, oz = torch.empty_like(u), torch.empty_like(u)
ck = torch.empty(B, Di // 64, L // 16, N, 64, device=dev)
scan_raw(u.transpose(1, 2), delta.transpose(1, 2), A, Bm.transpose(1, 2).unsqueeze(1), Cm.transpose(1, 2).unsqueeze(1), D, z.transpose(1, 2),
         db, True, out=out.transpose(1, 2), out_z=oz.transpose(1, 2), checkpoints=ck)
def timeit(fn, iters=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / iters * 1e3
print(f"probe mask {PROBE}:") if PROBE else None
print("scan bwd, own checkpoints     ", round(timeit(lambda: scan_bwd_tok(u, delta, A, Bm, Cm, D, z, db, dout, out, True)), 1), "us")
print("scan bwd, forward's checkpoints", round(timeit(lambda: scan_bwd_tok(u, delta, A, Bm, Cm, D, z, db, dout, out, True, checkpoints=ck)), 1), "us")
