"""conv_x_proj exactly as the model calls it at the headline shape, N launches (driver for the rocprofv3 PMC passes of tools/pmc_conv_xproj.sh)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zigma_amd.selective_scan_interface import conv_x_proj
dev, dt = "cuda", torch.bfloat16
B, L, Di, n = int(os.environ.get("B", 64)), int(os.environ.get("L", 1024)), 1280, 72
torch.manual_seed(0)
xz = torch.randn(B, L, 2 * Di, device=dev, dtype=dt)
cw = (0.5 * torch.randn(Di, 4, device=dev)).to(dt); cb = (0.5 * torch.randn(Di, device=dev)).to(dt)
w = (Di ** -0.5 * torch.randn(n, Di, device=dev)).to(dt)
perm = torch.randperm(L, device=dev).to(torch.int32)
for _ in range(int(os.environ.get("N", 20))):
    conv_x_proj(xz[:, :, :Di], cw, cb, w, perm)
torch.cuda.synchronize()
