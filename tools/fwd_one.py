"""Two forwards of the headline model (README model, bf16, B=64) — driver for per-kernel rocprofv3 PMC passes (tools/pmc_projections.sh)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
wl = bench.WORKLOADS["readme_text_b64"]
m = bench.build_model(wl["model"], "cuda", torch.bfloat16)
x, t, y = bench.make_inputs(wl, 64, "cuda", 0)
with torch.no_grad():
    for _ in range(int(os.environ.get("N", 2))):
        m(x, t, y)
torch.cuda.synchronize()
