"""When does every workgroup of the scan kernel start and finish?  (probe flag 0x1000: s_memtime stamps into the checkpoints buffer)"""
import json, os, sys, torch
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PROBE_LIB = os.path.join(ROOT, "tools", "libzigma_scan_probes.so")      # the stamps are compiled only into a probe build
if not os.path.exists(PROBE_LIB):               # (build it in the container before the GPU call)
    from zigma_amd import build as zbuild
    zbuild.build(verbose=False, lib=PROBE_LIB, extra_flags=("-DZIGMA_SCAN_PROBES",))
os.environ["ZIGMA_AMD_LIB"] = PROBE_LIB
from zigma_amd.selective_scan_interface import scan_raw
dev, dt = "cuda", torch.bfloat16
B, L, Di, N, R = 64, 1024, 1280, 16, 40
torch.manual_seed(0)
xz = torch.randn(B, L, 2 * Di, device=dev, dtype=dt); u = torch.randn(B, L, Di, device=dev, dtype=dt)
delta = (0.5 * torch.rand(B, L, Di, device=dev)).to(dt); xdbl = torch.randn(B, L, R + 2 * N, device=dev, dtype=dt)
A = -torch.exp(torch.log(torch.arange(1, N + 1, device=dev).float()) + 0.1 * torch.randn(Di, N, device=dev)).contiguous()
D = torch.randn(Di, device=dev); perm = torch.randperm(L, device=dev).to(torch.int32)
Bv = xdbl[:, :, R:R + N].transpose(1, 2).unsqueeze(1); Cv = xdbl[:, :, R + N:].transpose(1, 2).unsqueeze(1)
y = torch.empty(B, L, Di, device=dev, dtype=dt)
for prio in (0, 1):       # 0: shipped (rotation), 1: rotation off
    ck = torch.zeros(B * 20 * 4, device=dev, dtype=torch.float32)
    for _ in range(3):
        scan_raw(u.transpose(1, 2), delta.transpose(1, 2), A, Bv, Cv, D, xz[:, :, Di:].transpose(1, 2), None, False, out_z=y.transpose(1, 2),
                 z_row_index=perm, out_row_index=perm, want_out=False, checkpoints=ck, _probe_flags=0x1000 | (prio << 9))
    torch.cuda.synchronize()
    st = ck.view(torch.int64).cpu().numpy().reshape(-1, 2).astype(np.float64)
    t0 = st[:, 0].min()
    s, e = (st[:, 0] - t0), (st[:, 1] - t0)
    unit = e.max() / 270.0          # ticks per us if the kernel takes ~270 us
    q = lambda a: [round(float(np.percentile(a, p)) / e.max(), 3) for p in (0, 10, 25, 50, 75, 90, 100)]
    by_round = [round(float(e[k * 256:(k + 1) * 256].mean() / e.max()), 3) for k in range(5)]
    print(json.dumps(dict(prio=prio, start_pct=q(s), end_pct=q(e), mean_end_by_dispatch_round=by_round,
                          dur_pct=q(e - s), ticks_total=float(e.max()))))
