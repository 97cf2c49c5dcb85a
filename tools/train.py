#!/usr/bin/env python
"""Training loop of the reference's train_acc.py (train_acc.py:426-448: flow-matching loss, AdamW) on the MI355X-native
model — synthetic latents, no dataloader / EMA / wandb.  Forward AND backward run on the HIP kernels
(LayerNormFn, MambaInnerTokFn); the projections, attention and optimizer are torch.

    python tools/train.py --config '{...}' --batch 32 --steps 20
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 tools/train.py ...   # data parallel

Multi-GPU = plain data parallelism: one process per GPU, `--batch` samples per rank, gradients averaged by
DistributedDataParallel's bucketed all-reduce (RCCL over xGMI), which torch overlaps with the backward."""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", required=True)
    ap.add_argument("--batch", type=int, default=32, help="per-GPU batch")
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--lr", type=float, default=1e-4)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--bucket-mb", type=int, default=100, help="DDP bucket size: xGMI rings are per-link bound, few large buckets")
    args = ap.parse_args()

    from zigma_amd import sharded_sampling as ss
    from zigma_amd.model_zigma import ZigMa
    from zigma_amd.transport import create_transport
    if not torch.cuda.is_available():
        raise SystemExit("tools/train.py needs a GPU (the HIP path has no CPU fallback)")
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    rank, world, _ = ss.init_from_env(backend="nccl", device=device)
    cfg = json.loads(args.config)
    torch.manual_seed(0)
    model = ZigMa(device=device, dtype=torch.bfloat16 if args.dtype == "bf16" else torch.float32, **cfg).train()
    net = model
    if world > 1:
        net = torch.nn.parallel.DistributedDataParallel(model, device_ids=[local_rank], bucket_cap_mb=args.bucket_mb,
                                                        gradient_as_bucket_view=True)
    opt = torch.optim.AdamW(model.parameters(), lr=args.lr, fused=True)
    tr = create_transport()
    g = torch.Generator(device="cpu").manual_seed(1234 + rank)
    frames = cfg.get("video_frames", 0)
    shape = ((frames,) if frames else ()) + (cfg["in_channels"], cfg["img_dim"], cfg["img_dim"])
    x1 = torch.randn((args.batch,) + shape, generator=g).to(device)
    kw = {}
    if cfg.get("has_text"):
        kw["y"] = torch.rand(args.batch, cfg["n_context_token"], cfg["d_context"], generator=g).to(device, model.x_embedder.proj.weight.dtype)
    elif cfg.get("num_classes", -1) > 0:
        kw["y"] = torch.randint(0, cfg["num_classes"], (args.batch,), generator=g).to(device)

    def step():
        opt.zero_grad(set_to_none=True)
        loss = tr.training_losses(net, x1, kw)["loss"].mean()
        loss.backward()
        opt.step()
        return loss

    losses = []
    for _ in range(args.warmup):
        step()
    ss.fence(device, world)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        losses.append(step().detach())
    ss.fence(device, world)
    dt = (time.perf_counter() - t0) / args.steps
    if rank == 0:
        print(json.dumps(dict(world=world, per_gpu_batch=args.batch, ms_per_step=round(dt * 1e3, 2),
                              samples_per_s=round(world * args.batch / dt, 1), first_loss=round(float(losses[0]), 4),
                              last_loss=round(float(losses[-1]), 4), max_mem_GB=round(torch.cuda.max_memory_allocated() / 2 ** 30, 2))))
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
