"""Run the BASELINE.json configs 3-5 once on one MI355X and print one JSON line each (timings + sanity).  `--only 3|3y|4|5` runs one of them
   (for a kernel trace per config: tools/prof_configs.sh).
   config 3: CelebA-latent shape (4x32x32), zigzagN8, unconditional, 50-step fixed-grid Euler ODE sampling, B=64
   config 3y: the same sampling run on the model the reference's yaml ships (config/model/zigzag8_b1_pe2.yaml: E=768, depth 24)
   config v2: the same sampling run on the reference's bidirectional yaml (config/model/sweep2_b1_pe2.yaml: E=768, depth 24, scan_type v2 — two scans per layer)
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


def config3(dev, dt, E, depth, tag, scan_type="zigzagN8"):
    cfg = dict(in_channels=4, img_dim=32, embed_dim=E, depth=depth, patch_size=1, scan_type=scan_type, use_pe=2)
    m = bench.build_model(cfg, dev, dt)
    fn = Sampler(create_transport("Linear", "velocity")).sample_ode(sampling_method="euler", num_steps=50)
    z = torch.randn(64, 4, 32, 32, device=dev)
    with torch.no_grad():
        sec, traj = timed(lambda: fn(z, m.forward), 1, warm=1)
    print(json.dumps(dict(config=tag, what=f"50-step Euler ODE sampling (49 NFE), B=64, unconditional E={E} depth={depth} {scan_type}",
                          s_per_batch=sec, samples_per_s=64 / sec, ms_per_nfe=sec / 49 * 1e3, tokens_per_s=64 * 1024 * 49 / sec,
                          finite=bool(torch.isfinite(traj[-1]).all()), out_shape=list(traj.shape))), flush=True)


def main():
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default=None, choices=["3", "3y", "v2", "4", "5"])
    only = ap.parse_args().only
    dev, dt = "cuda", torch.bfloat16
    timer = bench.ScanTimer()
    timer.install()
    # ---- config 3 ------------------------------------------------------------------------------------------
    if only in (None, "3"):
        config3(dev, dt, 640, 18, 3)
    if only in (None, "3y"):      # the shipped yaml of BASELINE configs[2] (reference config/model/zigzag8_b1_pe2.yaml:7-8)
        config3(dev, dt, 768, 24, "3y")
    if only in (None, "v2"):      # reference config/model/sweep2_b1_pe2.yaml:4-10
        config3(dev, dt, 768, 24, "v2", scan_type="v2")
    if only in (None, "4"):
        config4(dev, dt, timer)
    if only in (None, "5"):
        config5(dev, dt)


def config4(dev, dt, timer):
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
    # VALU floor of the sequence-split formulation at the measured issue rates (profiles/r02_ubench2_valu_rates.txt: v_exp_f32 3.43 ns, packed 2.28,
    # plain 1.35 per wave-instruction per SIMD): the full pass costs 4 exp + 6 packed + 4 plain per (step, 4 states) = 32.8 ns, the state-only
    # first pass 4 exp + 4 plain mul + 4 plain mul + 4 fma = 4 * 3.43 + 12 * 1.35 = 29.9 ns — the recurrence is evaluated twice
    groups = B * L * Di * N / 64 / 4
    rp, rk, re_ = bench.measured_valu_rates()                      # profiles/valu_rates_gfx950.json
    floor_us = groups * ((4 * re_ + 6 * rk + 4 * rp) + (4 * re_ + 12 * rp)) * 1e-9 / 1024 * 1e6
    print(json.dumps(dict(config=4, what="L=16384 (4x128x128, patch 1), B=4, E=640 depth=18 zigzagN8: forward + scan kernel",
                          ms_per_forward=sec * 1e3, tokens_per_s=B * L / sec, scan_us=ms * 1e3, scan_algo_GBps=by / (ms * 1e-3) / 1e9,
                          scan_frac_of_8TBps=by / (ms * 1e-3) / 8e12, valu_floor_us=floor_us, valu_frac=floor_us / (ms * 1e3),
                          valu_floor_note="both passes of the sequence split at the measured VALU issue rates, 100 % pipe utilisation",
                          finite=bool(torch.isfinite(out).all()))), flush=True)


def config5(dev, dt):
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
