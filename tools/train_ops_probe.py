import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import WORKLOADS, build_model, make_inputs
from zigma_amd.transport import create_transport
from torch.profiler import profile, ProfilerActivity
dev = torch.device("cuda", 0)
wl = WORKLOADS["readme_text_b64"]
m = build_model(wl["model"], dev, torch.bfloat16).train()
opt = torch.optim.AdamW(m.parameters(), lr=1e-4, fused=True)
tr = create_transport()
x, t, y = make_inputs(wl, int(os.environ.get("B", 16)), dev, 0)
def step():
    opt.zero_grad(set_to_none=True)
    loss = tr.training_losses(m, x, dict(y=y))["loss"].mean()
    loss.backward(); opt.step()
step()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    step()
torch.cuda.synchronize()
ka = sorted(prof.key_averages(group_by_input_shape=True), key=lambda e: -e.device_time_total)
for e in ka:
    if not any(k in e.key for k in ("ndex", "ather", "catter", "take", "mbedding")): continue
    print(f"{e.key[:46]:46s} n={e.count:4d} dev_ms={e.device_time_total / 1e3:8.2f} shapes={str(e.input_shapes)[:110]}")

seen = set()
for e in prof.events():
    if any(k in e.name for k in ("index", "gather", "scatter", "take_along")) and e.name.startswith("aten::") and e.name not in seen:
        seen.add(e.name)
        print(e.name, e.input_shapes, [s_ for s_ in (e.stack or []) if "zigma" in s_ or "transport" in s_ or "bench" in s_][:5])
