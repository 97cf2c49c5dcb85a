"""Run the BASELINE.json configs 3-5 once on one MI355X and print one JSON line each (timings + sanity).
   config 3: CelebA-latent shape (4x32x32), zigzagN8, unconditional, 50-step fixed-grid Euler ODE sampling, B=64
   config 4: FacesHQ1024-latent shape (4x128x128, patch 1 -> L=16384), E=640 depth=18, B=4: forward + scan roofline
   config 5: UCF101 video (16 frames, 4x32x32, patch 2), zzvideo_sst, E=768 depth=24, 101 classes, B=2: forward"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from zigma_amd.transport import Sampler, create_transport


def timed(fn, iters, warm=1):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        out = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters, out


def main():
    dev, dt = "cuda", torch.bfloat16
    timer = bench.ScanTimer()
    timer.install()
    # ---- config 3 ------------------------------------------------------------------------------------------
    cfg = dict(in_channels=4, img_dim=32, embed_dim=640, depth=18, patch_size=1, scan_type="zigzagN8", use_pe=2)
    m = bench.build_model(cfg, dev, dt)
    fn = Sampler(create_transport("Linear", "velocity")).sample_ode(sampling_method="euler", num_steps=50)
    z = torch.randn(64, 4, 32, 32, device=dev)
    with torch.no_grad():
        sec, traj = timed(lambda: fn(z, m.forward), 1, warm=1)
    print(json.dumps(dict(config=3, what="50-step Euler ODE sampling (49 NFE), B=64, unconditional E=640 depth=18 zigzagN8",
                          s_per_batch=sec, samples_per_s=64 / sec, ms_per_nfe=sec / 49 * 1e3, tokens_per_s=64 * 1024 * 49 / sec,
                          finite=bool(torch.isfinite(traj[-1]).all()), out_shape=list(traj.shape))), flush=True)
    del m, traj
    # ---- config 4 ------------------------------------------------------------------------------------------
    cfg = dict(in_channels=4, img_dim=128, embed_dim=640, depth=18, patch_size=1, scan_type="zigzagN8", use_pe=2)
    m = bench.build_model(cfg, dev, dt)
    x, t = torch.randn(4, 4, 128, 128, device=dev), torch.rand(4, device=dev)
    timer.pairs.clear()
    timer.enabled = True
    with torch.no_grad():
        sec, out = timed(lambda: m(x, t), 3, warm=1)
    timer.enabled = False
    torch.cuda.synchronize()
    B, L, Di, N = 4, 16384, 1280, 16
    by = B * L * (4 * 2 * Di + 2 * 2 * N) + 4 * Di * (N + 2)
    ms = timer.mean_ms()
    print(json.dumps(dict(config=4, what="L=16384 (4x128x128, patch 1), B=4, E=640 depth=18 zigzagN8: forward + scan kernel",
                          ms_per_forward=sec * 1e3, tokens_per_s=B * L / sec, scan_us=ms * 1e3, scan_algo_GBps=by / (ms * 1e-3) / 1e9,
                          scan_frac_of_8TBps=by / (ms * 1e-3) / 8e12, finite=bool(torch.isfinite(out).all()))), flush=True)
    del m, out
    # ---- config 5 ------------------------------------------------------------------------------------------
    cfg = dict(in_channels=4, img_dim=32, embed_dim=768, depth=24, patch_size=2, num_classes=101, video_frames=16,
               scan_type="zzvideo_sst", use_pe=2)
    m = bench.build_model(cfg, dev, dt)
    x, t, y = torch.randn(2, 16, 4, 32, 32, device=dev), torch.rand(2, device=dev), torch.randint(0, 101, (2,), device=dev)
    from zigma_amd.graphs import GraphedForward
    with torch.no_grad():
        sec, out = timed(lambda: m(x, t, y), 3, warm=1)
        gf = GraphedForward(m, x, t, y)               # ~330 short launches: host-bound in eager mode, hipGraph removes that
        gsec, gout = timed(lambda: gf(x, t, y), 10, warm=2)
    print(json.dumps(dict(config=5, what="UCF101 video 16x(4x32x32) patch 2, zzvideo_sst, E=768 depth=24, B=2: forward",
                          ms_per_forward=sec * 1e3, tokens_per_s=2 * 4096 / sec, ms_per_forward_hipgraph=gsec * 1e3,
                          tokens_per_s_hipgraph=2 * 4096 / gsec, graph_matches_eager=bool(torch.equal(out, gout)),
                          finite=bool(torch.isfinite(out).all()), out_shape=list(out.shape))), flush=True)


if __name__ == "__main__":
    main()
