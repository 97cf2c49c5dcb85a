"""A/B at the headline shape (B=64, L=1024, d_inner=1280, n=72, bf16, zigzag row table): the one-pass conv + SiLU + x_proj kernel
against the two separate kernels (conv_tok, x_proj_mfma), interleaved rounds in one process.  Prints one JSON line."""
import json, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zigma_amd.causal_conv1d_interface import causal_conv1d_raw
from zigma_amd.selective_scan_interface import conv_x_proj, x_proj, dt_proj_softplus
dev, dt = "cuda", torch.bfloat16
B, L, Di, n = int(os.environ.get("B", 64)), int(os.environ.get("L", 1024)), 1280, 72
torch.manual_seed(0)
xz = torch.randn(B, L, 2 * Di, device=dev, dtype=dt)
cw = (0.5 * torch.randn(Di, 4, device=dev)).to(dt); cb = (0.5 * torch.randn(Di, device=dev)).to(dt)
w = (Di ** -0.5 * torch.randn(n, Di, device=dev)).to(dt)
perm = torch.randperm(L, device=dev).to(torch.int32)
R = 40
dw = (R ** -0.5 * torch.randn(Di, R, device=dev)).to(dt); db = torch.rand(Di, device=dev)
x_half = xz[:, :, :Di]
u_sep = torch.empty(B, L, Di, device=dev, dtype=dt)


def separate():
    causal_conv1d_raw(x_half.transpose(1, 2), cw, cb, True, out=u_sep.transpose(1, 2), x_row_index=perm)
    return u_sep, x_proj(u_sep, w)


def separate_dt():
    u, xd = separate()
    return u, xd, dt_proj_softplus(xd, R, dw, db, True)


def fused_then_dt():
    u, xd = conv_x_proj(x_half, cw, cb, w, perm)
    return u, xd, dt_proj_softplus(xd, R, dw, db, True)


F = lambda fl: (lambda: conv_x_proj(x_half, cw, cb, w, perm, _flags=fl))
variants = {"separate": separate, "separate_dt": separate_dt, "fused_then_dt": fused_then_dt,
            "fused": F(0), "fused_3stage": F(1), "fused_8w": F(2), "fused_8w_3stage": F(3),
            "probe_nostore": F(4), "probe_noconv": F(8), "probe_neither": F(12)}
outs = {k: f() for k, f in variants.items()}
torch.cuda.synchronize()
times = {k: [] for k in variants}
for rnd in range(6):
    for k, f in variants.items():
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            f()
        e1.record(); torch.cuda.synchronize()
        times[k].append(e0.elapsed_time(e1) / 10 * 1e3)
by = B * L * Di * 2 * 2 + B * L * n * 2
res = dict(shape=f"B={B} L={L} Di={Di} n={n} bf16", us_median={k: sorted(v)[len(v) // 2] for k, v in times.items()},
           us_min={k: min(v) for k, v in times.items()},
           hbm_frac_of_8TBps_fused={k: by / (sorted(times[k])[3] * 1e-6) / 8e12 for k in ("fused", "fused_3stage", "fused_8w", "fused_8w_3stage")},
           u_mismatch_frac=float((outs["fused"][0] != outs["separate"][0]).float().mean()),
           xdbl_max_abs_diff=float((outs["fused"][1].float() - outs["separate"][1].float()).abs().max()))
print(json.dumps(res))
