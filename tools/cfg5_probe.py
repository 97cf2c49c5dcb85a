import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
dev, dt = "cuda", torch.bfloat16
cfg = dict(in_channels=4, img_dim=32, embed_dim=768, depth=24, patch_size=2, num_classes=101, video_frames=16, scan_type="zzvideo_sst", use_pe=2)
m = bench.build_model(cfg, dev, dt)
x, t, y = torch.randn(2, 16, 4, 32, 32, device=dev), torch.rand(2, device=dev), torch.randint(0, 101, (2,), device=dev)
with torch.no_grad():
    for _ in range(5): m(x, t, y)
torch.cuda.synchronize()
