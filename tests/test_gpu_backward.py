"""GPU parity of the BACKWARD kernels through the C ABI.

fp32 I/O: gradients against autograd through the unmodified reference's pure-torch forward (tests/golden/bwd_*.npz,
oracle/make_golden_bwd.py) — the reference's own bound for its CUDA backward is rtol 6e-4 / atol 2e-3 on du, ddelta, dz
and 5x that on the reduced gradients (test_selective_scan.py:137-149); here: norm-wise <= 5e-5 on everything.
bf16 I/O: identical bf16 inputs on both sides, float64 oracle, bound norm-wise <= 1e-2 (bf16 outputs) / 2e-3 (f32 sums).
"""
import numpy as np
import pytest
import torch

from conftest import load_golden, rel_err
from oracle import bwd_cases
from oracle import zigma_oracle as zo

pytestmark = pytest.mark.gpu
DEV = "cuda"


def T(a, dtype=torch.float32):
    return None if a is None else torch.as_tensor(np.ascontiguousarray(a) if isinstance(a, np.ndarray) else a).to(DEV).to(dtype)


def N(t):
    return None if t is None else t.detach().float().cpu().numpy()


def tok(t):           # (B, D, L) -> token-major (B, L, D) contiguous
    return None if t is None else t.transpose(1, 2).contiguous()


def run_scan_bwd(c, dtype):
    from zigma_amd.selective_scan_interface import scan_bwd_tok, scan_raw
    u, delta, z, dout = (tok(T(c[k], dtype)) for k in ("u", "delta", "z", "dout"))
    Bm, Cm = tok(T(c["B"], dtype)), tok(T(c["C"], dtype))
    A, D, db = T(c["A"]), T(c["D"]), T(c["delta_bias"])
    out = None
    if z is not None:      # the forward's ungated y, from the forward kernel itself
        out = torch.empty_like(u)
        oz = torch.empty_like(u)
        scan_raw(u.transpose(1, 2), delta.transpose(1, 2), A, Bm.transpose(1, 2).unsqueeze(1), Cm.transpose(1, 2).unsqueeze(1),
                 D, z.transpose(1, 2), db, c["softplus"], out=out.transpose(1, 2), out_z=oz.transpose(1, 2))
    return scan_bwd_tok(u, delta, A, Bm, Cm, D, z, db, dout, out, c["softplus"])


@pytest.mark.parametrize("name", ["bwd_scan_full", "bwd_scan_plain", "bwd_scan_n8", "bwd_scan_long"])
def test_scan_bwd_fp32_vs_reference_autograd(name):
    from zigma_amd import _lib
    g = load_golden(name + ".npz")
    c = bwd_cases.scan_inputs(name)
    du, ddelta, dA, dB, dC, dD, dz, dbias = run_scan_bwd(c, torch.float32)
    assert _lib.last_kernel() == "scan_bwd_tok"
    L = c["u"].shape[-1]
    got = dict(du=du.transpose(1, 2), ddelta=ddelta.transpose(1, 2), dA=dA, dB=dB.transpose(1, 2), dC=dC.transpose(1, 2),
               dD=dD, dz=None if dz is None else dz.transpose(1, 2), ddelta_bias=dbias)
    for key, val in got.items():
        if key not in g:
            assert val is None
            continue
        v = N(val)
        if name == "bwd_scan_long" and v.shape[-1] == L:
            v = v[..., bwd_cases.LONG_KEEP]
        assert rel_err(v, g[key]) < 5e-5, (key, rel_err(v, g[key]))
        assert np.allclose(v, g[key], rtol=3e-3, atol=1e-2 if key in ("dA", "dD", "ddelta_bias") else 2e-3), key


@pytest.mark.parametrize("Bsz,Dm,L,Nst", [(2, 128, 64, 16), (1, 64, 37, 8)])
def test_scan_bwd_bf16_vs_oracle(Bsz, Dm, L, Nst):
    rng = np.random.default_rng(L)
    c = dict(u=zo.bf16_round(rng.standard_normal((Bsz, Dm, L)).astype(np.float32)),
             delta=zo.bf16_round((0.5 * rng.random((Bsz, Dm, L))).astype(np.float32)),
             A=(-0.5 * rng.random((Dm, Nst)) - 0.05).astype(np.float32),
             B=zo.bf16_round(rng.standard_normal((Bsz, Nst, L)).astype(np.float32)),
             C=zo.bf16_round(rng.standard_normal((Bsz, Nst, L)).astype(np.float32)),
             D=rng.standard_normal(Dm).astype(np.float32), z=zo.bf16_round(rng.standard_normal((Bsz, Dm, L)).astype(np.float32)),
             delta_bias=(0.5 * rng.random(Dm)).astype(np.float32),
             dout=zo.bf16_round(rng.standard_normal((Bsz, Dm, L)).astype(np.float32)), softplus=True)
    du, ddelta, dA, dB, dC, dD, dz, dbias = run_scan_bwd(c, torch.bfloat16)
    ref = zo.selective_scan_bwd(c["u"], c["delta"], c["A"], c["B"], c["C"], c["D"], c["z"], c["delta_bias"], c["dout"], True)
    assert rel_err(N(du.transpose(1, 2)), ref["du"]) < 1e-2
    assert rel_err(N(ddelta.transpose(1, 2)), ref["ddelta"]) < 1e-2
    assert rel_err(N(dz.transpose(1, 2)), ref["dz"]) < 1e-2        # uses the bf16-rounded forward y
    for got, key in ((dA, "dA"), (dB.transpose(1, 2), "dB"), (dC.transpose(1, 2), "dC"), (dD, "dD"), (dbias, "ddelta_bias")):
        assert rel_err(N(got), ref[key]) < 2e-3, key


def test_scan_bwd_is_bit_reproducible_and_linear_in_dout():
    """Full-size property checks (B=8, L=1024, Di=1280): two runs are bit-identical (no atomics); the backward is
    linear in dout for fixed inputs."""
    from zigma_amd.selective_scan_interface import scan_bwd_tok
    g = torch.Generator(device="cpu").manual_seed(0)
    Bsz, L, Dm, Nst = 8, 1024, 1280, 16
    r = lambda *s: torch.randn(*s, generator=g).to(DEV)
    u, delta, dout1, dout2 = r(Bsz, L, Dm), (0.5 * torch.rand(Bsz, L, Dm, generator=g)).to(DEV), r(Bsz, L, Dm), r(Bsz, L, Dm)
    A = (-0.5 * torch.rand(Dm, Nst, generator=g) - 0.05).to(DEV)
    Bm, Cm, D = r(Bsz, L, Nst), r(Bsz, L, Nst), r(Dm)
    run = lambda do: scan_bwd_tok(u, delta, A, Bm, Cm, D, None, None, do, None, True)
    a, b2 = run(dout1), run(dout1)
    for x, y in zip(a, b2):
        assert (x is None and y is None) or torch.equal(x, y)
    c, s = run(dout2), run(dout1 + dout2)
    for x, y, w in zip(a, c, s):
        if x is not None:
            assert rel_err(N(x + y), N(w)) < 1e-5
    assert all(torch.isfinite(t).all() for t in a if t is not None)
