"""Round-6 same-process A/Bs (interleaved rounds, one box): module knobs flipped between timed forwards of the same model object.
  config 4   (B=4, L=16384)          selective_scan_interface.DT_PROJ_IN_SPLIT  on | off   (dt_proj + softplus inside the split's first pass | dt_proj kernel)
  B = 8      README model, hipGraph   the same knob
  v2         (E=768, depth 24, B=64)  selective_scan_interface.ACCUMULATE_IN_SCAN on | off  (reversed sweep adds in its scan epilogue | in-place add pass)
  config 5   (video, B=2, hipGraph)   in-kernel dt_proj with reset_period on | off (dt_in_scan_eligible patched to refuse reset_period)
Prints one JSON line per case; also written to gpurun_out/r06_ab.jsonl."""
import json, os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
import zigma_amd.selective_scan_interface as ssi
from zigma_amd.graphs import GraphedForward
dev, dt = "cuda", torch.bfloat16
out = open(os.path.join(ROOT, "gpurun_out", "r06_ab.jsonl"), "w") if os.path.isdir(os.path.join(ROOT, "gpurun_out")) else None


def timed(fn, n):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def ab(tag, build, setters, n, rounds=4, graph=False):
    """setters: {name: fn(on: bool)}; build() -> (model, args)"""
    m, args = build()
    res = {}
    with torch.no_grad():
        fns = {}
        for name, setter in setters.items():
            setter()
            fns[name] = GraphedForward(m, *args) if graph else None       # (a graph captures the routing in force when it is recorded)
        for r in range(rounds):
            for name, setter in setters.items():
                setter()
                f = (lambda: fns[name](*args)) if graph else (lambda: m(*args))
                res.setdefault(name, []).append(timed(f, n))
    line = dict(case=tag, graph=graph, ms={k: [round(x, 3) for x in v] for k, v in res.items()}, ms_median={k: sorted(v)[len(v) // 2] for k, v in res.items()})
    print(json.dumps(line), flush=True)
    if out:
        out.write(json.dumps(line) + "\n"); out.flush()
    del m
    torch.cuda.empty_cache()


def knob(mod, name, val):
    return lambda: setattr(mod, name, val)


which = sys.argv[1:] or ["4", "b8", "v2", "5"]
if "4" in which:
    cfg = dict(in_channels=4, img_dim=128, embed_dim=640, depth=18, patch_size=1, scan_type="zigzagN8", use_pe=2)
    ab("config4_B4_L16384", lambda: (bench.build_model(cfg, dev, dt), (torch.randn(4, 4, 128, 128, device=dev), torch.rand(4, device=dev))),
       {"dt_in_split": knob(ssi, "DT_PROJ_IN_SPLIT", True), "dt_proj_kernel": knob(ssi, "DT_PROJ_IN_SPLIT", False)}, n=5)
    ssi.DT_PROJ_IN_SPLIT = True
if "b8" in which:
    wl = bench.WORKLOADS["readme_text_b64"]
    ab("readme_B8_hipgraph", lambda: (bench.build_model(wl["model"], dev, dt), bench.make_inputs(wl, 8, dev, 0)),
       {"dt_in_split": knob(ssi, "DT_PROJ_IN_SPLIT", True), "dt_proj_kernel": knob(ssi, "DT_PROJ_IN_SPLIT", False)}, n=20, graph=True)
    ssi.DT_PROJ_IN_SPLIT = True
if "v2" in which:
    cfg = dict(in_channels=4, img_dim=32, embed_dim=768, depth=24, patch_size=1, scan_type="v2", use_pe=2)
    ab("sweep2_v2_E768_B64", lambda: (bench.build_model(cfg, dev, dt), (torch.randn(64, 4, 32, 32, device=dev), torch.rand(64, device=dev))),
       {"add_in_scan": knob(ssi, "ACCUMULATE_IN_SCAN", True), "add_pass": knob(ssi, "ACCUMULATE_IN_SCAN", False)}, n=5)
    ssi.ACCUMULATE_IN_SCAN = True
if "5" in which:
    cfg = dict(in_channels=4, img_dim=32, embed_dim=768, depth=24, patch_size=2, num_classes=101, video_frames=16, scan_type="zzvideo_sst", use_pe=2)
    real = ssi.dt_in_scan_eligible
    def no_reset(on):
        def f():
            ssi.dt_in_scan_eligible = real if on else (lambda u, x, w, reset_period=0, *a, **k: False if reset_period else real(u, x, w, reset_period, *a, **k))
        return f
    ab("config5_video_B2_hipgraph", lambda: (bench.build_model(cfg, dev, dt), (torch.randn(2, 16, 4, 32, 32, device=dev), torch.rand(2, device=dev), torch.randint(0, 101, (2,), device=dev))),
       {"dt_in_scan_with_reset": no_reset(True), "dt_proj_kernel_for_temporal": no_reset(False)}, n=20, graph=True)
    ssi.dt_in_scan_eligible = real
