"""Training-step timing of the README model (bf16, AdamW, flow-matching loss) through the HIP forward + backward kernels."""
import json, os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import WORKLOADS, build_model, make_inputs
from zigma_amd.transport import create_transport
import zigma_amd.wgrad as _wg
_wg.SPLIT_WGRAD = os.environ.get("SPLIT_WGRAD", "1") == "1"      # A/B knob
dev = torch.device("cuda", 0)
wl = WORKLOADS["readme_text_b64"]
B = int(os.environ.get("B", 32))
m = build_model(wl["model"], dev, torch.bfloat16).train()
opt = torch.optim.AdamW(m.parameters(), lr=1e-4, fused=True)
tr = create_transport()
x, t, y = make_inputs(wl, B, dev, 0)
def step():
    opt.zero_grad(set_to_none=True)
    loss = tr.training_losses(m, x, dict(y=y))["loss"].mean()
    loss.backward()
    opt.step()
    return loss
for _ in range(2): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
n = 5
for _ in range(n): l = step()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
print(json.dumps(dict(what="train step (fwd+bwd+AdamW), README model bf16", batch=B, ms_per_step=dt * 1e3, tokens_per_s=B * 1024 / dt,
                      loss=float(l.detach()), max_mem_GB=torch.cuda.max_memory_allocated() / 2**30)))
