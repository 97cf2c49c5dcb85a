"""Standalone timing of the HIP kernels at the BASELINE config-2 shapes (B=64, L=1024, E=640, Di=1280, N=16, bf16).
Prints one JSON line per kernel with achieved algorithmic GB/s.  GPU only."""
import json
import sys
import os

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zigma_amd import _lib
from zigma_amd.causal_conv1d_interface import causal_conv1d_raw
from zigma_amd.layernorm import block_norm
from zigma_amd.selective_scan_interface import scan_raw


def timeit(fn, iters=30, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def main():
    dev, dt = "cuda", torch.bfloat16
    B, L, E, N, R = int(os.environ.get("B", 64)), int(os.environ.get("L", 1024)), 640, 16, 40
    Di = 2 * E
    torch.manual_seed(0)
    xz = torch.randn(B, L, 2 * Di, device=dev, dtype=dt)
    u = torch.randn(B, L, Di, device=dev, dtype=dt)
    delta = (torch.rand(B, L, Di, device=dev) - 0.5).to(dt)
    xdbl = torch.randn(B, L, R + 2 * N, device=dev, dtype=dt)
    A = -torch.exp(torch.log(torch.arange(1, N + 1, device=dev).float())).repeat(Di, 1).contiguous()
    D, db = torch.randn(Di, device=dev), torch.rand(Di, device=dev) * 0.5
    perm = torch.randperm(L, device=dev).to(torch.int32)
    y = torch.empty(B, L, Di, device=dev, dtype=dt)
    Bv = xdbl[:, :, R:R + N].transpose(1, 2).unsqueeze(1)
    Cv = xdbl[:, :, R + N:].transpose(1, 2).unsqueeze(1)

    def scan():
        scan_raw(u.transpose(1, 2), delta.transpose(1, 2), A, Bv, Cv, D, xz[:, :, Di:].transpose(1, 2), db, True,
                 out_z=y.transpose(1, 2), z_row_index=perm, out_row_index=perm, want_out=False)
    t = timeit(scan)
    by = B * L * (4 * 2 * Di + 2 * 2 * N) + 4 * Di * (N + 2)
    print(json.dumps(dict(kernel=_lib.last_kernel(), us=t * 1e6, algo_GBps=by / t / 1e9, frac_of_8TBps=by / t / 8e12)))

    if os.environ.get("SCAN_ONLY"):
        return
    w, bias = torch.randn(Di, 4, device=dev, dtype=dt), torch.randn(Di, device=dev, dtype=dt)
    uo = torch.empty(B, L, Di, device=dev, dtype=dt)

    def conv():
        causal_conv1d_raw(xz[:, :, :Di].transpose(1, 2), w, bias, True, out=uo.transpose(1, 2), x_row_index=perm)
    t = timeit(conv)
    by = B * L * Di * 2 * 2
    print(json.dumps(dict(kernel=_lib.last_kernel(), us=t * 1e6, algo_GBps=by / t / 1e9, frac_of_8TBps=by / t / 8e12)))

    x = torch.randn(B, L, E, device=dev, dtype=dt)
    br = torch.randn(B, L, E, device=dev, dtype=dt)
    res = torch.randn(B, L, E, device=dev)
    mod = torch.randn(B, 6 * E, device=dev, dtype=dt)
    wn = torch.ones(E, device=dev, dtype=dt)

    def norm():
        block_norm(x, wn, None, res, 1e-5, True, branch=br, gate=mod[:, 2 * E:3 * E], shift=mod[:, :E], scale=mod[:, E:2 * E])
    t = timeit(norm)
    by = B * L * E * (2 + 2 + 4 + 4 + 2 + 2)     # x, branch, res in, res out, n, xm
    print(json.dumps(dict(kernel=_lib.last_kernel(), us=t * 1e6, algo_GBps=by / t / 1e9, frac_of_8TBps=by / t / 8e12)))

    from zigma_amd.selective_scan_interface import dt_proj_softplus
    Wdt48 = torch.zeros(Di, 48, device=dev, dtype=dt)[:, :R]
    Wdt48.copy_(torch.randn(Di, R, device=dev, dtype=dt) * 0.1)
    t = timeit(lambda: dt_proj_softplus(xdbl, R, Wdt48, db, True))
    by = B * L * (Di * 2 + (R + 2 * N) * 2)
    print(json.dumps(dict(kernel=_lib.last_kernel(), us=t * 1e6, algo_GBps=by / t / 1e9, frac_of_8TBps=by / t / 8e12)))

    # library GEMMs of one block, for the whole-forward budget
    Win = torch.randn(2 * Di, E, device=dev, dtype=dt)
    Wx = torch.randn(R + 2 * N, Di, device=dev, dtype=dt)
    Wdt = torch.randn(Di, R, device=dev, dtype=dt)
    Wout = torch.randn(E, Di, device=dev, dtype=dt)
    F = torch.nn.functional
    for name, fn, fl in (("in_proj", lambda: F.linear(x, Win), 2 * B * L * E * 2 * Di),
                         ("x_proj", lambda: F.linear(u, Wx), 2 * B * L * Di * (R + 2 * N)),
                         ("dt_proj", lambda: F.linear(xdbl[:, :, :R], Wdt), 2 * B * L * R * Di),
                         ("out_proj", lambda: F.linear(y, Wout), 2 * B * L * Di * E)):
        t = timeit(fn)
        print(json.dumps(dict(kernel="torch " + name, us=t * 1e6, TFLOPs=fl / t / 1e12)))


if __name__ == "__main__":
    main()
