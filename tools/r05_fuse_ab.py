"""Same-process A/B of the headline forward (README model, B=64, bf16) with module-level knobs toggled between interleaved rounds:
out_proj's gated add fused into the 4-wave kernel's epilogue (default) vs the plain 4-wave product + the add inside the next norm kernel."""
import json, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
import zigma_amd.model_zigma as mz
import zigma_amd.mamba_simple as zms
wl = bench.WORKLOADS["readme_text_b64"]
dev = torch.device("cuda", 0)
model = bench.build_model(wl["model"], dev, torch.bfloat16)
x, t, y = bench.make_inputs(wl, wl["batch"], dev, seed=1234)


def run(n):
    with torch.no_grad():
        for _ in range(n):
            out = model(x, t, y)
    return out


variants = {"default": {}, "out_proj_add_unfused": {(mz, "FUSE_OUT_PROJ_ADD"): False}}
extra = os.environ.get("EXTRA")
outs, times = {}, {k: [] for k in variants}
for rnd in range(5):
    for name, knobs in variants.items():
        saved = {k: getattr(*k) for k in knobs}
        for k, v in knobs.items():
            setattr(*k, v)
        outs[name] = run(2)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); run(10); e1.record(); torch.cuda.synchronize()
        times[name].append(e0.elapsed_time(e1) / 10)
        for k, v in saved.items():
            setattr(*k, v)
res = {"ms_median": {k: round(sorted(v)[len(v) // 2], 3) for k, v in times.items()}, "ms_all": {k: [round(a, 3) for a in v] for k, v in times.items()},
       "rel_diff_vs_default": {k: float((outs[k].float() - outs["default"].float()).norm() / outs["default"].float().norm()) for k in variants}}
print(json.dumps(res))
json.dump(res, open(os.path.join(ROOT, "gpurun_out", "r05_fuse_ab.json"), "w"), indent=1)
