import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import WORKLOADS, build_model, make_inputs
import zigma_amd.selective_scan_interface as ssi
dev = torch.device("cuda", 0)
wl = WORKLOADS["readme_text_b64"]
m = build_model(wl["model"], dev, torch.bfloat16)
x, t, y = make_inputs(wl, 64, dev, 0)
def timeit(fn, n=15):
    for _ in range(4): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
with torch.no_grad():
    for rep in range(2):
        for flag in (False, True):
            ssi.USE_X_PROJ_KERNEL = flag
            print("x_proj kernel" if flag else "library x_proj", round(timeit(lambda: m(x, t, y)), 2), "ms", flush=True)
