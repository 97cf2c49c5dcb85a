"""Batch-sharded sampling across the GPUs of one node.

Mirrors how the reference's `sample_acc.py` scales out: one process per GPU, the global batch split evenly
over ranks, weights replicated, per-rank RNG seed = global_seed + rank (sample_acc.py:58), NO collective inside
the denoiser forward or the ODE loop, one gather of the finished latents at the end (accelerator.gather,
sample_acc.py:435) and barriers around timed regions (accelerator.wait_for_everyone).  Here that is plain
`torch.distributed`: backend "nccl" (= RCCL over xGMI on ROCm) on GPUs, "gloo" in the CPU tests.
"""
import os
import time

import torch
import torch.distributed as dist


def init_from_env(backend=None, device=None):
    """(rank, world, local_rank) from the torchrun environment; initialises the process group when world > 1."""
    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        backend = backend or ("nccl" if torch.cuda.is_available() else "gloo")
        kw = {}
        if backend == "nccl" and device is not None:
            kw["device_id"] = device
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return rank, world, local_rank


def rank_seed(global_seed, rank):
    return int(global_seed) + int(rank)


def local_batch(global_batch, rank, world):
    """Even split; the reference asserts divisibility too (sample_acc.py:275-277)."""
    if global_batch % world != 0:
        raise ValueError(f"global batch {global_batch} must be divisible by world size {world}")
    return global_batch // world


def gather_samples(x, world, out=None):
    """all_gather of the per-rank samples along dim 0 (rank order), like accelerator.gather."""
    if world == 1:
        return x
    x = x.contiguous()
    if out is None:
        out = torch.empty((world * x.shape[0],) + tuple(x.shape[1:]), device=x.device, dtype=x.dtype)
    dist.all_gather_into_tensor(out, x)
    return out


def fence(device, world):
    if device is not None and torch.device(device).type == "cuda":
        torch.cuda.synchronize(device)
    if world > 1:
        dist.barrier()
    if device is not None and torch.device(device).type == "cuda":
        torch.cuda.synchronize(device)


def timed_steps(step_fn, steps, warmup, device, world):
    """W untimed warm-up steps, then exactly K steps bracketed by barrier + synchronize on both sides;
    returns the MAX over ranks of the elapsed seconds."""
    for _ in range(warmup):
        step_fn()
    fence(device, world)
    t0 = time.perf_counter()
    for _ in range(steps):
        step_fn()
    fence(device, world)
    elapsed = time.perf_counter() - t0
    if world > 1:
        el = torch.tensor([elapsed], dtype=torch.float64, device=device if torch.device(device).type == "cuda" else "cpu")
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
        elapsed = float(el.item())
    return elapsed


def sample_sharded(sample_fn, model_fn, shape, global_batch, global_seed, device, dtype=torch.float32, **model_kwargs):
    """Draw this rank's share of the initial noise with the per-rank seed, integrate with `sample_fn`
    (a `Sampler.sample_ode(...)` closure) and gather the final latents from all ranks."""
    rank = dist.get_rank() if dist.is_initialized() else 0
    world = dist.get_world_size() if dist.is_initialized() else 1
    nb = local_batch(global_batch, rank, world)
    g = torch.Generator(device="cpu").manual_seed(rank_seed(global_seed, rank))
    z = torch.randn((nb,) + tuple(shape), generator=g, dtype=dtype).to(device)
    final = sample_fn(z, model_fn, **model_kwargs)[-1]
    return gather_samples(final, world)
