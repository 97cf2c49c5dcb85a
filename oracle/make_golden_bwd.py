"""Golden gradients for the backward of the hot path, from autograd through the UNMODIFIED reference's pure-torch
forward functions (selective_scan_ref, causal_conv1d_ref, rms_norm_ref / layer_norm_ref, mamba_inner_ref) — the same
thing the reference's own tests compare its CUDA backward against (tests/ops/test_selective_scan.py:100-149).

TEST INFRASTRUCTURE ONLY (build container; needs /root/reference).
    python -m oracle.make_golden_bwd   ->  tests/golden/bwd_*.npz
"""
import contextlib
import io
import os

import numpy as np
import torch

import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))     # repo root: `python -m oracle.make_golden_bwd` or `python oracle/make_golden_bwd.py`
from oracle import bwd_cases, ref_shim  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def npy(t):
    return None if t is None else t.detach().to(torch.float32).numpy()


def save(name, **arrs):
    arrs = {k: (v if isinstance(v, np.ndarray) else npy(v)) for k, v in arrs.items() if v is not None}
    np.savez_compressed(os.path.join(OUT, name), **arrs)
    print(name, {k: v.shape for k, v in arrs.items()})


def main():
    with contextlib.redirect_stdout(io.StringIO()):
        mz, ssi, cci, uz = ref_shim.reference_modules()
    f64 = torch.float64                       # conv / norm goldens: autograd in double
    f32 = torch.float32                       # selective_scan_ref / mamba_inner_ref cast their operands to float themselves

    # ---- selective scan (inputs come from oracle/bwd_cases.py seeds and are NOT stored) -----------------------
    for name in bwd_cases.SCAN_CASES:
        c = bwd_cases.scan_inputs(name)
        leaves = {k: v for k, v in c.items() if torch.is_tensor(v) and k != "dout"}
        for t in leaves.values():
            t.requires_grad_(True)
        out = ssi.selective_scan_ref(c["u"], c["delta"], c["A"], c["B"], c["C"], c["D"], c["z"], c["delta_bias"],
                                     c["softplus"])
        out.backward(c["dout"])
        grads = {"d" + ("delta_bias" if k == "delta_bias" else k): v.grad for k, v in leaves.items()}
        grads["out"] = out
        if name == "bwd_scan_long":           # keep the fixture small: time-indexed tensors at LONG_KEEP positions only
            keep = torch.tensor(bwd_cases.LONG_KEEP)
            grads = {k: (v.index_select(-1, keep) if v.shape[-1] == c["u"].shape[-1] else v) for k, v in grads.items()}
        save(name + ".npz", **grads)

    # ---- causal conv1d ------------------------------------------------------------------------------
    for name, (Bsz, Dm, L, W, act, seed) in {"bwd_conv_silu.npz": (2, 64, 50, 4, "silu", 6),
                                             "bwd_conv_plain.npz": (2, 12, 7, 3, None, 7)}.items():
        g = torch.Generator().manual_seed(seed)
        x = torch.randn(Bsz, Dm, L, generator=g, dtype=f64, requires_grad=True)
        w = torch.randn(Dm, W, generator=g, dtype=f64, requires_grad=True)
        b = torch.randn(Dm, generator=g, dtype=f64, requires_grad=True)
        dout = torch.randn(Bsz, Dm, L, generator=g, dtype=f64)
        out = cci.causal_conv1d_ref(x, w, b, activation=act)
        out.backward(dout)
        save(name, x=x, weight=w, bias=b, dout=dout, out=out, dx=x.grad, dweight=w.grad, dbias=b.grad,
             silu=np.array(int(act is not None)))

    # ---- add + norm (prenorm, residual in fp32 -> here everything in double) ------------------------------
    ln = ref_shim.reference_layernorm()
    for name, (rows, cols, rms, has_res, seed) in {"bwd_norm_rms.npz": ((2, 9), 64, True, True, 8),
                                                   "bwd_norm_ln.npz": ((3, 5), 96, False, True, 9),
                                                   "bwd_norm_rms_nores.npz": ((4,), 640, True, False, 10)}.items():
        g = torch.Generator().manual_seed(seed)
        x = torch.randn(*rows, cols, generator=g, dtype=f64, requires_grad=True)
        res = torch.randn(*rows, cols, generator=g, dtype=f64, requires_grad=True) if has_res else None
        w = (1 + 0.1 * torch.randn(cols, generator=g, dtype=f64)).requires_grad_(True)
        b = None if rms else (0.1 * torch.randn(cols, generator=g, dtype=f64)).requires_grad_(True)
        dy = torch.randn(*rows, cols, generator=g, dtype=f64)
        dres = torch.randn(*rows, cols, generator=g, dtype=f64)
        fn = ln.rms_norm_ref if rms else ln.layer_norm_ref
        y, res_out = fn(x, w, b, residual=res, eps=1e-5, prenorm=True, upcast=False)
        torch.autograd.backward([y, res_out], [dy, dres])
        save(name, x=x, residual=res, weight=w, bias=b, dy=dy, dresidual_out=dres, y=y, dx=x.grad,
             dresidual=None if res is None else res.grad, dweight=w.grad, dbias=None if b is None else b.grad,
             rms=np.array(int(rms)), eps=np.array(1e-5))

    # ---- mamba inner (conv -> x_proj -> dt_proj -> scan -> out_proj), all parameter gradients ---------------
    g = torch.Generator().manual_seed(11)
    Bsz, E, Di, L, N, R, W = 2, 32, 64, 48, 16, 8, 4
    r = lambda *s: torch.randn(*s, generator=g, dtype=f32)
    leaves = dict(xz=r(Bsz, 2 * Di, L), conv_w=r(Di, 1, W) * 0.5, conv_b=r(Di) * 0.1, x_proj_w=r(R + 2 * N, Di) * Di ** -0.5,
                  dt_proj_w=r(Di, R) * R ** -0.5, out_proj_w=r(E, Di) * Di ** -0.5, out_proj_b=r(E) * 0.1,
                  A=-torch.exp(r(Di, N) * 0.5), D=r(Di), delta_bias=torch.rand(Di, generator=g, dtype=f32) * 0.5)
    for t in leaves.values():
        t.requires_grad_(True)
    # mamba_inner_ref routes through the extension-backed autograd Functions; for the golden the two ops are the
    # reference's own pure-torch refs (identical forward math, gradients by autograd)
    ssi.selective_scan_fn = ssi.selective_scan_ref
    ssi.causal_conv1d_fn = lambda x, w, b, act: cci.causal_conv1d_ref(x, w, b, activation=act)
    out = ssi.mamba_inner_ref(leaves["xz"], leaves["conv_w"], leaves["conv_b"], leaves["x_proj_w"], leaves["dt_proj_w"],
                              leaves["out_proj_w"], leaves["out_proj_b"], leaves["A"], None, None, leaves["D"],
                              delta_bias=leaves["delta_bias"], delta_softplus=True)
    dout = r(*out.shape)
    out.backward(dout)
    save("bwd_mamba_inner.npz", dout=dout, out=out, **leaves, **{"d_" + k: v.grad for k, v in leaves.items()})


def model_grads():
    """Parameter / input gradients of the whole (small) reference model: the reference's Mamba calls `mamba_inner_fn`
    (an autograd Function over its CUDA forward/backward); here that name is bound to the reference's own pure-torch
    `mamba_inner_ref` with pure-torch conv / scan inside, so the gradients are autograd's."""
    import ast
    with contextlib.redirect_stdout(io.StringIO()):
        mz, ssi, cci, uz = ref_shim.reference_modules()
    import dis_mamba.mamba_ssm.modules.mamba_simple as ms
    ssi.selective_scan_fn = ssi.selective_scan_ref
    ssi.causal_conv1d_fn = lambda x, w, b, act: cci.causal_conv1d_ref(x, w, b, activation=act)
    ms.mamba_inner_fn = ssi.mamba_inner_ref
    for name in ("zigma_text_zigzag2", "zigma_uncond_zigzag8"):
        g = np.load(os.path.join(OUT, name + ".npz"))
        cfg = ast.literal_eval(str(g["cfg"]))
        with contextlib.redirect_stdout(io.StringIO()):
            m = mz.ZigMa(device="cpu", **cfg).eval()
        m.load_state_dict({k[3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("sd.")})
        x = torch.from_numpy(g["x"]).requires_grad_(True)
        t = torch.from_numpy(g["t"])
        y = torch.from_numpy(g["y"]) if "y" in g.files else None
        out = m(x, t, y)
        assert np.allclose(out.detach().numpy(), g["out"], rtol=1e-4, atol=1e-5), "differentiable path != fixture forward"
        wgt = torch.randn(out.shape, generator=torch.Generator().manual_seed(77))
        (out * wgt).sum().backward()
        arrs = {"g." + k: p.grad for k, p in m.named_parameters() if p.grad is not None}
        arrs["gx"] = x.grad
        arrs["wgt"] = wgt
        save("bwd_model_" + name + ".npz", **arrs)


if __name__ == "__main__":
    import sys
    if "--model-only" not in sys.argv:
        main()
    model_grads()
