"""Same-process A/B of the bench forward (README text model, bf16, B=64) with and without the four small kernels of zigma_amd/embed.py
(patch embedding, timestep features, skinny projections, final layer), interleaved rounds.  Prints one JSON line."""
import json, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import WORKLOADS, build_model, make_inputs
from zigma_amd import embed
dev = torch.device("cuda", 0)
wl = WORKLOADS["readme_text_b64"]
m = build_model(wl["model"], dev, torch.bfloat16).eval()
x, t, y = make_inputs(wl, 64, dev, 0)
def run(flag, n=10):
    embed.USE_EMBED_KERNELS = flag
    with torch.no_grad():
        for _ in range(2): m(x, t, y)
        torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): out = m(x, t, y)
        e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n, out
res = {True: [], False: []}
outs = {}
for rnd in range(4):
    for flag in (False, True):
        ms, outs[flag] = run(flag)
        res[flag].append(ms)
embed.USE_EMBED_KERNELS = True
d = (outs[True].float() - outs[False].float())
print(json.dumps(dict(what="README text model bf16 B=64 forward", ms_torch_composition=sorted(res[False])[1], ms_own_kernels=sorted(res[True])[1],
                      all_rounds=res, rel_diff=float(d.norm() / outs[False].float().norm()))))
