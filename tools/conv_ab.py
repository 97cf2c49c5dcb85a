"""A/B of the token-major conv kernels at the headline shape (B=64, L=1024, Di=1280, bf16, zigzag gather): 8-byte accesses
(ZIGMA_CONV_V1=1) against the 16-byte form, interleaved; plus dt_proj."""
import json, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zigma_amd.causal_conv1d_interface import causal_conv1d_raw
from zigma_amd.selective_scan_interface import dt_proj_softplus
dev, dt = "cuda", torch.bfloat16
B, L, Di, R = 64, 1024, 1280, 40
torch.manual_seed(0)
xz = torch.randn(B, L, 2 * Di, device=dev, dtype=dt)
w, bias = torch.randn(Di, 4, device=dev, dtype=dt), torch.randn(Di, device=dev, dtype=dt)
perm = torch.randperm(L, device=dev).to(torch.int32)
outs = {}
def run(name, v1):
    if v1: os.environ["ZIGMA_CONV_V1"] = "1"
    else: os.environ.pop("ZIGMA_CONV_V1", None)
    o = outs.setdefault(name, torch.empty(B, L, Di, device=dev, dtype=dt))
    causal_conv1d_raw(xz[:, :, :Di].transpose(1, 2), w, bias, True, out=o.transpose(1, 2), x_row_index=perm)
times = {"v1_8B": [], "v2_16B": []}
for rnd in range(6):
    for name, v1 in (("v1_8B", True), ("v2_16B", False)):
        run(name, v1); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): run(name, v1)
        e1.record(); torch.cuda.synchronize()
        times[name].append(e0.elapsed_time(e1) / 10 * 1e3)
xdbl = torch.randn(B, L, 72, device=dev, dtype=dt)
Wdt = torch.zeros(Di, 48, device=dev, dtype=dt)[:, :R]; Wdt.copy_(torch.randn(Di, R, device=dev, dtype=dt) * 0.1)
db = torch.rand(Di, device=dev)
for _ in range(3): dt_proj_softplus(xdbl, R, Wdt, db, True)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): dt_proj_softplus(xdbl, R, Wdt, db, True)
e1.record(); torch.cuda.synchronize()
by = B * L * Di * 2 * 2
med = {k: sorted(v)[len(v) // 2] for k, v in times.items()}
print(json.dumps(dict(conv_us=med, conv_TBps={k: by / (v * 1e-6) / 1e12 for k, v in med.items()}, identical=bool(torch.equal(outs["v1_8B"], outs["v2_16B"])),
                      dt_proj_us=e0.elapsed_time(e1) / 20 * 1e3)))
