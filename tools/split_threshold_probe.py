import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import WORKLOADS, build_model, make_inputs
import zigma_amd.selective_scan_interface as ssi
dev = torch.device("cuda", 0)
wl = WORKLOADS["readme_text_b64"]
m = build_model(wl["model"], dev, torch.bfloat16)
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
for B in (8, 12, 16, 24, 32):
    x, t, y = make_inputs(wl, B, dev, 0)
    res = []
    for thr in (0, 256, 384, 512, 700):
        ssi.SPLIT_MAX_WGS = thr
        with torch.no_grad():
            res.append((thr, round(timeit(lambda: m(x, t, y)), 2)))
    print(B, B * 20, res, flush=True)
