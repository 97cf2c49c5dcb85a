"""in_conv_x_proj (x half of in_proj + conv + SiLU + x_proj in one kernel) against the two-kernel path it replaces, at the headline
shape: parity of u and x_dbl, then interleaved timings with HIP events.  PROBES=1 adds the phase probes (results wrong)."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from zigma_amd import scan_paths                                                             # noqa: E402
from zigma_amd.linear import linear                                                          # noqa: E402
from zigma_amd.selective_scan_interface import conv_x_proj, in_conv_x_proj                   # noqa: E402
import torch.nn.functional as F                                                              # noqa: E402


def timeit(fn, n=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for a, b in ev:
        a.record()
        fn()
        b.record()
    torch.cuda.synchronize()
    t = sorted(a.elapsed_time(b) * 1e3 for a, b in ev)
    return t[len(t) // 2]


def main():
    out = {}
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    shapes = ((64, 1024, 640, 1280, 72, "zigzag"), (4, 16384, 640, 1280, 72, "zigzag"), (32, 1024, 640, 1280, 80, "none"))
    if os.environ.get("ONE") == "1":
        shapes = shapes[:1]
    for (B, L, E, Di, n, tab) in shapes:
        h = torch.randn(B, L, E, device=dev).bfloat16()
        w_in = (torch.randn(2 * Di, E, device=dev) * E ** -0.5).bfloat16()
        conv_w = (torch.randn(Di, 4, device=dev) * 0.5).bfloat16()
        conv_b = (torch.randn(Di, device=dev) * 0.5).bfloat16()
        w_x = (torch.randn(n, Di, device=dev) * Di ** -0.5).bfloat16()
        perm = None
        if tab == "zigzag":
            side = int(L ** 0.5)
            perm = torch.as_tensor(scan_paths.zigzag_path(side)[1].copy(), dtype=torch.int32, device=dev)
        xz = F.linear(h, w_in)
        u_ref, xd_ref = conv_x_proj(xz[:, :, :Di], conv_w, conv_b, w_x, perm)
        u, xd = in_conv_x_proj(h, w_in[:Di], conv_w, conv_b, w_x, perm)
        torch.cuda.synchronize()
        du = (u.float() - u_ref.float())
        dx = (xd.float() - xd_ref.float())
        key = f"B{B}_L{L}_E{E}"
        out[key] = {"u_rel": (du.norm() / u_ref.float().norm()).item(), "u_max": du.abs().max().item(),
                    "u_frac_diff": (du != 0).float().mean().item(),
                    "xdbl_rel": (dx.norm() / xd_ref.float().norm()).item(), "xdbl_max": dx.abs().max().item()}
        # where do the differences sit (position inside the 128-tile)?
        bad = (du.abs() > 0.05).any(dim=2)
        if bad.any():
            idx = bad.nonzero()[:12].tolist()
            out[key]["bad_positions"] = idx
            out[key]["bad_count"] = int(bad.sum())
        t_lin = timeit(lambda: F.linear(h, w_in))
        t_linz = timeit(lambda: F.linear(h, w_in[Di:]))
        t_own_z = timeit(lambda: linear(h, w_in[Di:]))
        t_own_zs = timeit(lambda: linear(h, w_in[Di:], silu_from_col=0))
        t_cx = timeit(lambda: conv_x_proj(xz[:, :, :Di], conv_w, conv_b, w_x, perm))
        t_icx = timeit(lambda: in_conv_x_proj(h, w_in[:Di], conv_w, conv_b, w_x, perm))
        out[key].update({"in_proj_lib_us": t_lin, "in_proj_z_lib_us": t_linz, "in_proj_z_own_us": t_own_z, "in_proj_z_own_silu_us": t_own_zs,
                         "conv_x_proj_us": t_cx, "in_conv_x_proj_us": t_icx,
                         "before_us": t_lin + t_cx, "after_us": min(t_linz, t_own_z) + t_icx})
        if os.environ.get("PROBES") == "1" and E == 640:
            for fl in (2, 4, 8, 12, 16, 30, 32, 62):
                out[key][f"probe_{fl}_us"] = timeit(lambda: in_conv_x_proj(h, w_in[:Di], conv_w, conv_b, w_x, perm, _flags=fl))
        print(key, json.dumps(out[key]), flush=True)
        os.makedirs("gpurun_out", exist_ok=True)
        json.dump(out, open("gpurun_out/icx_probe.json", "w"), indent=1)


if __name__ == "__main__":
    main()
