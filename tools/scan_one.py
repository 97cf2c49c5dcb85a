"""The scan exactly as the model calls it at the headline shape, N launches (driver for rocprofv3 PMC passes: tools/pmc_scan.sh)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zigma_amd.selective_scan_interface import scan_raw
dev, dt = "cuda", torch.bfloat16
B, L, Di, N, R = int(os.environ.get("B", 64)), int(os.environ.get("L", 1024)), 1280, 16, 40
torch.manual_seed(0)
xz = torch.randn(B, L, 2 * Di, device=dev, dtype=dt); u = torch.randn(B, L, Di, device=dev, dtype=dt)
delta = (0.5 * torch.rand(B, L, Di, device=dev)).to(dt); xdbl = torch.randn(B, L, R + 2 * N, device=dev, dtype=dt)
A = -torch.exp(torch.log(torch.arange(1, N + 1, device=dev).float()) + 0.1 * torch.randn(Di, N, device=dev)).contiguous()
D = torch.randn(Di, device=dev); perm = torch.randperm(L, device=dev).to(torch.int32)
y = torch.empty(B, L, Di, device=dev, dtype=dt)
Bv = xdbl[:, :, R:R + N].transpose(1, 2).unsqueeze(1); Cv = xdbl[:, :, R + N:].transpose(1, 2).unsqueeze(1)
from zigma_amd import selective_scan_interface as ssi
dt_in = ssi.DT_PROJ_IN_SCAN and os.environ.get("DT_IN", "1") == "1"      # as the model launches it since round 4: dt_proj + softplus inside
w_dt = (R ** -0.5 * torch.randn(Di, R, device=dev)).to(dt); db = torch.randn(Di, device=dev) - 3
for _ in range(int(os.environ.get("N", 20))):
    if dt_in:
        scan_raw(u.transpose(1, 2), None, A, Bv, Cv, D, xz[:, :, Di:].transpose(1, 2), db, True,
                 out_z=y.transpose(1, 2), z_row_index=perm, out_row_index=perm, want_out=False, dt_x=xdbl, dt_w=w_dt)
    else:
        scan_raw(u.transpose(1, 2), delta.transpose(1, 2), A, Bv, Cv, D, xz[:, :, Di:].transpose(1, 2), None, False,
                 out_z=y.transpose(1, 2), z_row_index=perm, out_row_index=perm, want_out=False)
torch.cuda.synchronize()
