"""to_q + attention core as one kernel (zigma_q_attn_fwd) against the pieces it replaces — the library to_q GEMM + cross_attn_kernel,
and the own projection kernel + cross_attn_kernel — at the headline shape (B=64, L=1024, E=640, 8 heads x 64, 77 keys, bf16).
Interleaved timing; one JSON line."""
import json, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zigma_amd.attention import cross_attn, q_attn, transpose_v
from zigma_amd.linear import linear
F = torch.nn.functional
dev, dt = "cuda", torch.bfloat16
B, L, E, H, NC = int(os.environ.get("B", 64)), 1024, 640, 8, 77
torch.manual_seed(0)
x = torch.randn(B, L, E, device=dev, dtype=dt); wq = (E ** -0.5 * torch.randn(H * 64, E, device=dev)).to(dt)
kv = torch.randn(B, NC, 2, H * 64, device=dev, dtype=dt); k, v = kv[:, :, 0], kv[:, :, 1]
vt = transpose_v(v)
variants = {"lib_to_q_then_attn": lambda: cross_attn(F.linear(x, wq), k, v, H), "own_to_q_then_attn": lambda: cross_attn(linear(x, wq, None), k, v, H),
            "q_attn": lambda: q_attn(x, wq, k, vt, H), "lib_to_q_only": lambda: F.linear(x, wq), "transpose_v": lambda: transpose_v(v)}
outs = {n: f() for n, f in variants.items()}
torch.cuda.synchronize()
times = {n: [] for n in variants}
for rnd in range(6):
    for n, f in variants.items():
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): f()
        e1.record(); torch.cuda.synchronize()
        times[n].append(e0.elapsed_time(e1) / 10 * 1e3)
a, b = outs["q_attn"].float(), outs["lib_to_q_then_attn"].float()
print(json.dumps(dict(shape=f"B={B} L={L} E={E} heads={H} n_ctx={NC} bf16", us_median={n: sorted(v)[len(v) // 2] for n, v in times.items()},
                      rel_diff_vs_two_kernels=float((a - b).norm() / b.norm()))))
