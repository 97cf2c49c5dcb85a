"""Warm vs cold operands: the projections of the block stand-alone with their input resident in L2 / Infinity Cache (back-to-back launches
over the same buffers, what the probes of tools/ time) against the same launch after 1 GB of unrelated traffic (what the forward gives them:
an input the previous kernel has just written, everything else evicted)."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from zigma_amd.linear import linear
dev, dt = "cuda", torch.bfloat16
M = 65536
junk = torch.empty(1 << 29, device=dev, dtype=torch.uint8)
junk2 = torch.empty(1 << 29, device=dev, dtype=torch.uint8)


def t_one(fn, cold, producer=None):
    ts = []
    for _ in range(12):
        if cold:
            junk2.copy_(junk)                      # 512 MB read + 512 MB written
        if producer is not None:
            producer()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts = sorted(ts[2:])
    return ts[len(ts) // 2]


res = torch.randn(64, 1024, 640, device=dev).to(dt)
gate = torch.randn(64, 640, device=dev).to(dt)
for name, K, N, kw in (("in_proj_ws", 640, 2560, dict(weight_stationary=True)), ("to_q", 640, 512, {}), ("out_proj+add", 1280, 640, dict(residual=res, gate=gate)),
                       ("to_out+bias+add", 512, 640, dict(residual=res, gate=gate, bias=True))):
    x = torch.randn(64, 1024, K, device=dev).to(dt)
    src = torch.randn(64, 1024, K, device=dev).to(dt)
    w = (torch.randn(N, K, device=dev) * K ** -0.5).to(dt)
    kw = dict(kw)
    if kw.pop("bias", False):
        kw["bias"] = torch.randn(N, device=dev).to(dt)
    xx = x if "residual" in kw else x.view(-1, K)
    fn = lambda: linear(xx, w, **kw)
    rec = dict(shape=f"{name} K={K} N={N}", warm_us=t_one(fn, False), cold_us=t_one(fn, True),
               cold_input_just_written_us=t_one(fn, True, producer=lambda: x.copy_(src)))
    print(json.dumps(rec), flush=True)
