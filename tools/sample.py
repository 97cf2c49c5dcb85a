#!/usr/bin/env python
"""Latent sampling with the MI355X-native ZigMa — the denoiser / ODE part of the reference's sample_acc.py
(sample_acc.py:147-176,300-449) without its VAE decode, metrics and wandb.

    python tools/sample.py --config '{"in_channels":4,"img_dim":32,"embed_dim":640,"depth":18,"scan_type":"zigzagN8","use_pe":2}' \
        [--ckpt model.pt] --num-samples 256 --batch 64 --steps 50 --method euler --out samples.pt
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 tools/sample.py ...   # batch-sharded

One process per GPU, the global batch split over the ranks, per-rank seed = seed + rank, no collective inside the ODE loop,
one all_gather of the finished latents per batch (zigma_amd/sharded_sampling.py).  `--graph` replays the denoiser as a
hipGraph (worth it for small per-GPU batches, where eager launches are host-bound).  Checkpoints of the reference load
unchanged ("ema" / "model" / plain state_dict)."""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", required=True, help="JSON of the ZigMa constructor arguments (the reference's config/model/*.yaml params)")
    ap.add_argument("--ckpt", default=None)
    ap.add_argument("--num-samples", type=int, default=64)
    ap.add_argument("--batch", type=int, default=64, help="GLOBAL batch per sampling call")
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--method", default="euler", help="euler | midpoint | heun2 | heun3 | rk4 | dopri5 | bosh3 | adaptive_heun")
    ap.add_argument("--path", default="Linear")
    ap.add_argument("--prediction", default="velocity")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--class-label", type=int, default=None, help="class-conditional models: the label to sample")
    ap.add_argument("--graph", action="store_true")
    ap.add_argument("--decode", default="none", choices=["none", "standin"],
                    help="standin: decode -> uint8 on the device inside the timed path (zigma_amd.postprocess; a 4->3 channel 8x "
                         "upsampling stand-in for the reference's third-party VAE, sample_acc.py:363-392) and gather the pixels")
    ap.add_argument("--out", default=None)
    args = ap.parse_args()

    from zigma_amd import sharded_sampling as ss
    from zigma_amd.graphs import GraphedForward
    from zigma_amd.model_zigma import ZigMa
    from zigma_amd.transport import Sampler, create_transport
    if not torch.cuda.is_available():
        raise SystemExit("tools/sample.py needs a GPU (the HIP path has no CPU fallback)")
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    rank, world, _ = ss.init_from_env(backend="nccl", device=device)
    cfg = json.loads(args.config)
    dtype = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    model = ZigMa(device=device, dtype=dtype, **cfg).eval()
    if args.ckpt:
        sd = torch.load(args.ckpt, map_location="cpu")
        sd = sd.get("ema", sd.get("model", sd)) if isinstance(sd, dict) else sd
        model.load_state_dict(sd)
    frames = cfg.get("video_frames", 0)
    shape = ((frames,) if frames else ()) + (cfg["in_channels"], cfg["img_dim"], cfg["img_dim"])
    nb = ss.local_batch(args.batch, rank, world)
    kw = {}
    if args.class_label is not None:
        kw["y"] = torch.full((nb,), args.class_label, device=device, dtype=torch.long)
    sample_fn = Sampler(create_transport(args.path, args.prediction)).sample_ode(sampling_method=args.method, num_steps=args.steps)
    model_fn = model.forward
    if args.graph:
        z0 = torch.zeros((nb,) + shape, device=device)
        model_fn = GraphedForward(model, z0, torch.zeros(nb, device=device), kw.get("y"))
    dec = None
    if args.decode == "standin":
        from zigma_amd import postprocess as pp
        dec = pp.StandInDecoder(latent_channels=cfg["in_channels"], device=device)
    outs, t0 = [], time.perf_counter()
    with torch.no_grad():
        for i in range(-(-args.num_samples // args.batch)):
            if dec is None:
                outs.append(ss.sample_sharded(sample_fn, model_fn, shape, args.batch, args.seed + i * world, device, **kw).cpu())
            else:       # this rank's latents -> images -> uint8 -> ONE gather of the pixels (sample_acc.py:362-392,435)
                g = torch.Generator(device="cpu").manual_seed(ss.rank_seed(args.seed + i * world, rank))
                z = torch.randn((nb,) + tuple(shape), generator=g).to(device)
                outs.append(pp.finish_samples(sample_fn(z, model_fn, **kw)[-1], dec, is_video=bool(frames), world=world).cpu())
    ss.fence(device, world)
    dt = time.perf_counter() - t0
    if rank == 0:
        x = torch.cat(outs)[:args.num_samples]
        print(json.dumps(dict(samples=int(x.shape[0]), shape=list(x.shape[1:]), seconds=round(dt, 3),
                              samples_per_s=round(x.shape[0] / dt, 2), world=world, decode=args.decode,
                              finite=bool(torch.isfinite(x.float()).all()))))
        if args.out:
            torch.save(x, args.out)
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
