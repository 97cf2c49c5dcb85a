"""s_memtime stamps of workgroup 1 of in_conv_x_proj (probe bit 128): per wave, the time between its meetings."""
import os, sys, json, torch, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zigma_amd import _lib
dev, dt = "cuda", torch.bfloat16
B, L, E, Di, n = 64, 1024, 640, 1280, 72
torch.manual_seed(0)
h = torch.randn(B, L, E, device=dev, dtype=dt)
w_in = (E ** -0.5 * torch.randn(Di, E, device=dev)).to(dt)
cw = (0.5 * torch.randn(Di, 4, device=dev)).to(dt); cb = (0.5 * torch.randn(Di, device=dev)).to(dt)
w = (Di ** -0.5 * torch.randn(n, Di, device=dev)).to(dt)
perm = torch.randperm(L, device=dev).to(torch.int32)
u = torch.empty(B, L, Di, device=dev, dtype=dt); xd = torch.empty(B, L, n, device=dev, dtype=dt)
P = _lib.InConvXProjParams()
P.batch, P.seqlen, P.dim, P.n, P.k, P.dtype, P.flags = B, L, Di, n, E, _lib.dtype_id(h), int(os.environ.get('FLAGS', 128))
P.h_batch_stride, P.h_l_stride = h.stride(0), h.stride(1)
P.u_batch_stride, P.u_l_stride = u.stride(0), u.stride(1)
P.win_row_stride, P.w_row_stride, P.out_row_stride = w_in.stride(0), w.stride(0), n
P.h, P.w_in, P.conv_weight, P.conv_bias, P.w = h.data_ptr(), w_in.data_ptr(), cw.data_ptr(), cb.data_ptr(), w.data_ptr()
P.u, P.out, P.x_row_index = u.data_ptr(), xd.data_ptr(), perm.data_ptr()
nb = _lib.lib().zigma_in_conv_x_proj_fwd_workspace_bytes(C.byref(P))
ws = torch.zeros(nb, device=dev, dtype=torch.uint8)
P.workspace, P.workspace_bytes = ws.data_ptr(), nb
for _ in range(3):
    _lib.call("zigma_in_conv_x_proj_fwd", P, torch.device(dev))
torch.cuda.synchronize()
st = ws[nb - 8 * 1024 * 8:].view(torch.int64).view(8, 1024).cpu()
out = {}
for wv in range(8):
    t = st[wv]; t = t[t > 0].tolist()
    out[f"wave{wv}"] = [x - t[0] for x in t]
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open(f"gpurun_out/icx_stamps_{os.environ.get('FLAGS', 128)}.json", "w"))
for wv in (0, 4):
    t = out[f"wave{wv}"]
    print("wave", wv, "n", len(t), "total", t[-1] if t else None)
    print(" first 40 deltas:", [t[i + 1] - t[i] for i in range(min(40, len(t) - 1))])
    mid = len(t) // 2
    print(" mid 30 deltas:", [t[i + 1] - t[i] for i in range(mid, min(mid + 30, len(t) - 1))])
