"""world_size-2 gloo tests of the batch-sharded sampling path (CPU; the GPU path is the same code on RCCL)."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from zigma_amd import sharded_sampling as ss
    from zigma_amd.transport import Sampler, create_transport
    r, w, _ = ss.init_from_env(backend="gloo")
    assert (r, w) == (rank, world)
    # 1) sharded ODE sampling with a per-sample-independent toy velocity field: rank order + per-rank seeds
    fn = Sampler(create_transport()).sample_ode(sampling_method="euler", num_steps=5)
    model = lambda x, t: -x * t.view(-1, 1, 1, 1)
    out = ss.sample_sharded(fn, model, (2, 3, 3), global_batch=6, global_seed=100, device="cpu")
    # reference: what a single process would compute for each rank's shard
    ref = []
    for rr in range(world):
        g = torch.Generator().manual_seed(100 + rr)
        z = torch.randn(3, 2, 3, 3, generator=g)
        ref.append(fn(z, model)[-1])
    ok_gather = torch.allclose(out, torch.cat(ref))
    # 2) timing harness: max over ranks, exactly K timed steps
    calls = []
    import time
    el = ss.timed_steps(lambda: (calls.append(1), time.sleep(0.02 * (rank + 1))), steps=3, warmup=2, device="cpu", world=world)
    # 3) the step after the ODE (sample_acc.py:363-392,435): decode -> uint8 -> gather in rank order
    from zigma_amd import postprocess as pp
    img = pp.finish_samples(torch.full((2, 3, 4, 4), float(rank)), decode=lambda z: z * pp.LATENT_SCALE - 0.5)
    ok_img = img.dtype == torch.uint8 and img.shape == (4, 3, 4, 4) and \
        torch.equal(img[:2], pp.to_uint8(torch.full((2, 3, 4, 4), -0.5))) and torch.equal(img[2:], pp.to_uint8(torch.full((2, 3, 4, 4), 0.5)))
    q.put((rank, ok_gather and ok_img, len(calls), el, tuple(out.shape)))
    dist.destroy_process_group()


def test_sharded_sampling_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank, ok_gather, ncalls, el, shape in res:
        assert ok_gather and shape == (6, 2, 3, 3)
        assert ncalls == 5                      # 2 warm-up + 3 timed
    els = [r[3] for r in res]
    assert abs(els[0] - els[1]) < 1e-9          # both ranks report the same (max) time
    assert els[0] >= 3 * 0.04 * 0.9             # the slower rank (2 x 20 ms per step) bounds it


def test_local_batch_and_seed_rules():
    sys.path.insert(0, ROOT)
    from zigma_amd import sharded_sampling as ss
    assert ss.local_batch(64, 3, 8) == 8 and ss.rank_seed(7, 3) == 10
    with pytest.raises(ValueError):
        ss.local_batch(10, 0, 4)


def test_bench_refuses_a_multi_gpu_run_it_cannot_launch():
    """`python bench.py --gpus N` spawns N ranks itself (torch.distributed.run); with fewer visible GPUs it must fail loudly
    instead of reporting a smaller job."""
    sys.path.insert(0, ROOT)
    import bench
    have = torch.cuda.device_count()
    with pytest.raises(SystemExit) as e:
        bench.respawn_under_launcher(have + 2)
    assert "refusing" in str(e.value)
