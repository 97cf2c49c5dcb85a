"""Small-batch (serving) latency of the README model: eager vs hipGraph replay."""
import json, os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import WORKLOADS, build_model, make_inputs
from zigma_amd.graphs import GraphedForward
dev = torch.device("cuda", 0)
wl = WORKLOADS["readme_text_b64"]
m = build_model(wl["model"], dev, torch.bfloat16)
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
for B in (1, 2, 4, 8, 16):
    x, t, y = make_inputs(wl, B, dev, 0)
    with torch.no_grad():
        e = timeit(lambda: m(x, t, y))
        gf = GraphedForward(m, x, t, y)
        g = timeit(lambda: gf(x, t, y))
        import zigma_amd.selective_scan_interface as ssi
        ssi.SPLIT_SMALL_BATCH = False
        g0 = timeit(GraphedForward(m, x, t, y).__call__ if False else (lambda gf0=GraphedForward(m, x, t, y): gf0(x, t, y)))
        ssi.SPLIT_SMALL_BATCH = True
    print(json.dumps(dict(batch=B, eager_ms=round(e, 2), hipgraph_ms=round(g, 2), hipgraph_ms_without_sequence_split=round(g0, 2),
                          tokens_per_s_hipgraph=round(B * 1024 / g * 1e3))), flush=True)
