import json, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import zigma_amd.model_zigma as mz
dev, dt = torch.device("cuda", 0), torch.bfloat16
cfg = dict(in_channels=4, img_dim=32, embed_dim=640, depth=18, patch_size=1, scan_type="zigzagN8", use_pe=2)
m = bench.build_model(cfg, dev, dt).eval()
x, t = torch.randn(64, 4, 32, 32, device=dev), torch.rand(64, device=dev)
def run(flag, n=20):
    mz.FUSE_OUT_PROJ_ADD_NO_TEXT = flag
    with torch.no_grad():
        for _ in range(3): m(x, t)
        torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): o = m(x, t)
        e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n, o
r = {True: [], False: []}
o = {}
for _ in range(3):
    for f in (False, True):
        ms, o[f] = run(f); r[f].append(ms)
print(json.dumps(dict(what="config 3 model (unconditional E=640 depth=18) B=64 forward, out_proj gated add fused (own kernel) vs library + add in the next norm",
                      ms_library=r[False], ms_fused=r[True], rel_diff=float((o[True].float() - o[False].float()).norm() / o[False].float().norm()))))
