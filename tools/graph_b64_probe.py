"""Eager vs hipGraph replay of the bench forward (README text model, bf16, B=64): how much of the step is launch gaps?"""
import json, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import WORKLOADS, build_model, make_inputs
from zigma_amd.graphs import GraphedForward
dev = torch.device("cuda", 0)
wl = WORKLOADS["readme_text_b64"]
B = int(os.environ.get("B", 64))
m = build_model(wl["model"], dev, torch.bfloat16).eval()
x, t, y = make_inputs(wl, B, dev, 0)
def timeit(fn, n=20):
    for _ in range(5): fn()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n
with torch.no_grad():
    eager = timeit(lambda: m(x, t, y))
    gf = GraphedForward(m, x, t, y)
    replay = timeit(lambda: gf.graph.replay())
    full = timeit(lambda: gf(x, t, y))
    same = bool(torch.equal(gf(x, t, y), m(x, t, y)))
print(json.dumps(dict(what="README text model bf16 forward", batch=B, eager_ms=eager, graph_replay_ms=replay, graph_call_with_copies_ms=full, identical=same)))
