"""BASELINE config 5 (UCF101 video, zzvideo_sst, E=768, depth 24, B=2) forward, N times — driver for a rocprofv3 kernel trace."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
cfg = dict(in_channels=4, img_dim=32, embed_dim=768, depth=24, patch_size=2, num_classes=101, video_frames=16,
           scan_type="zzvideo_sst", use_pe=2)
m = bench.build_model(cfg, "cuda", torch.bfloat16)
x, t, y = torch.randn(2, 16, 4, 32, 32, device="cuda"), torch.rand(2, device="cuda"), torch.randint(0, 101, (2,), device="cuda")
with torch.no_grad():
    for _ in range(int(os.environ.get("N", 8))):
        m(x, t, y)
torch.cuda.synchronize()
