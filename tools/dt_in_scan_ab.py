"""dt_proj inside the scan (ABI 9, zigma_scan_params_t.dt_x) against dt_proj_softplus_kernel + scan at the headline shape:
interleaved HIP-event timings of (a) dt_proj kernel, (b) scan on the bf16 delta tensor, (c) scan with the product inside."""
import json, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zigma_amd.selective_scan_interface import dt_proj_softplus, scan_raw
dev, dt = "cuda", torch.bfloat16
B, L, Di, N, R = int(os.environ.get("B", 64)), int(os.environ.get("L", 1024)), 1280, 16, 40
torch.manual_seed(0)
xz = torch.randn(B, L, 2 * Di, device=dev, dtype=dt); u = torch.randn(B, L, Di, device=dev, dtype=dt)
xdbl = torch.randn(B, L, R + 2 * N, device=dev, dtype=dt)
w = (R ** -0.5 * torch.randn(Di, R, device=dev)).to(dt); db = torch.randn(Di, device=dev) - 3
A = -torch.exp(torch.log(torch.arange(1, N + 1, device=dev).float()) + 0.1 * torch.randn(Di, N, device=dev)).contiguous()
D = torch.randn(Di, device=dev); perm = torch.randperm(L, device=dev).to(torch.int32)
y = torch.empty(B, L, Di, device=dev, dtype=dt)
Bv = xdbl[:, :, R:R + N].transpose(1, 2).unsqueeze(1); Cv = xdbl[:, :, R + N:].transpose(1, 2).unsqueeze(1)
z = xz[:, :, Di:].transpose(1, 2)
delta = dt_proj_softplus(xdbl, R, w, db, True)
f_dt = lambda: dt_proj_softplus(xdbl, R, w, db, True)
f_scan = lambda: scan_raw(u.transpose(1, 2), delta.transpose(1, 2), A, Bv, Cv, D, z, None, False, out_z=y.transpose(1, 2),
                          z_row_index=perm, out_row_index=perm, want_out=False)
f_in = lambda: scan_raw(u.transpose(1, 2), None, A, Bv, Cv, D, z, db, True, out_z=y.transpose(1, 2), z_row_index=perm,
                        out_row_index=perm, want_out=False, dt_x=xdbl, dt_w=w)
fs = {"dt_proj_kernel": f_dt, "scan_on_delta": f_scan, "scan_dt_in_kernel": f_in}
for f in fs.values():
    f()
torch.cuda.synchronize()
times = {k: [] for k in fs}
for rnd in range(8):
    for k, f in fs.items():
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            f()
        e1.record(); torch.cuda.synchronize()
        times[k].append(e0.elapsed_time(e1) / 10 * 1e3)
med = {k: sorted(v)[len(v) // 2] for k, v in times.items()}
algo_old = B * L * (4 * 2 * Di + 2 * 2 * N) + 4 * Di * (N + 2)
algo_new = B * L * (3 * 2 * Di + 2 * (R + 2 * N)) + 4 * Di * (N + 2) + 2 * Di * R
res = dict(shape=f"B={B} L={L} Di={Di} N={N} R={R} bf16", us_median=med, us_min={k: min(v) for k, v in times.items()},
           two_kernels_us=med["dt_proj_kernel"] + med["scan_on_delta"],
           algorithmic_bytes=dict(scan_on_delta=algo_old, scan_dt_in_kernel=algo_new),
           hbm_frac=dict(scan_on_delta=algo_old / (med["scan_on_delta"] * 1e-6) / 8e12, scan_dt_in_kernel=algo_new / (med["scan_dt_in_kernel"] * 1e-6) / 8e12,
                         pair_old=(algo_old + B * L * Di * 2 + B * L * R * 2) / ((med["dt_proj_kernel"] + med["scan_on_delta"]) * 1e-6) / 8e12))
print(json.dumps(res))
os.makedirs("gpurun_out", exist_ok=True)
json.dump(res, open("gpurun_out/dt_in_scan_ab.json", "w"), indent=1)
