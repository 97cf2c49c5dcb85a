"""in_conv_x_proj at the headline shape, N launches (driver for rocprofv3 PMC passes, tools/pmc_icx.sh)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zigma_amd.selective_scan_interface import in_conv_x_proj
dev, dt = "cuda", torch.bfloat16
B, L, E, Di, n = int(os.environ.get("B", 64)), int(os.environ.get("L", 1024)), 640, 1280, 72
torch.manual_seed(0)
h = torch.randn(B, L, E, device=dev, dtype=dt)
w_in = (E ** -0.5 * torch.randn(Di, E, device=dev)).to(dt)
cw = (0.5 * torch.randn(Di, 4, device=dev)).to(dt); cb = (0.5 * torch.randn(Di, device=dev)).to(dt)
w = (Di ** -0.5 * torch.randn(n, Di, device=dev)).to(dt)
perm = torch.randperm(L, device=dev).to(torch.int32)
for _ in range(int(os.environ.get("N", 10))):
    in_conv_x_proj(h, w_in, cw, cb, w, perm, _flags=int(os.environ.get("FLAGS", 0)))
torch.cuda.synchronize()
