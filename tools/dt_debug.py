import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zigma_amd.selective_scan_interface import scan_raw
import torch.nn.functional as F
dev = "cuda"
Bsz, L, Di, R, N = 1, 32, 64, 40, 16
torch.manual_seed(0)
u = torch.randn(Bsz, L, Di, device=dev).bfloat16(); z = torch.randn(Bsz, L, Di, device=dev).bfloat16()
xd = torch.randn(Bsz, L, R + 2 * N, device=dev).bfloat16(); w = (torch.randn(Di, R, device=dev) * R ** -0.5).bfloat16()
A = -torch.rand(Di, N, device=dev) - 0.5; D = torch.randn(Di, device=dev); db = torch.randn(Di, device=dev) - 3
Bv, Cv = xd[:, :, R:R + N].transpose(1, 2).unsqueeze(1), xd[:, :, R + N:].transpose(1, 2).unsqueeze(1)
y = torch.empty(Bsz, L, Di, device=dev, dtype=torch.bfloat16)
ck = torch.full((Bsz, L, Di, 2), -777.0, device=dev)
scan_raw(u.transpose(1, 2), None, A, Bv, Cv, D, z.transpose(1, 2), db, True, out_z=y.transpose(1, 2), want_out=False, dt_x=xd, dt_w=w, checkpoints=ck)
torch.cuda.synchronize()
ref = F.softplus(xd[:, :, :R].float() @ w.float().t() + db)
d = (ck[..., 0] - ref).abs()
print("dt max err by step", [round(v, 4) for v in d[0].max(1).values.tolist()])
du = (ck[..., 1] - u.float()).abs()
print("u' max err by step", [round(v, 4) for v in du[0].max(1).values.tolist()])
print("unwritten", (ck == -777).float().mean().item())
