"""The numpy backward restatements (oracle/zigma_oracle.py) against gradients obtained by autograd through the
UNMODIFIED reference's pure-torch forward (oracle/make_golden_bwd.py -> tests/golden/bwd_*.npz).  CPU only."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import bwd_cases  # noqa: E402
from oracle import zigma_oracle as zo  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30)


@pytest.mark.parametrize("name", list(bwd_cases.SCAN_CASES))
def test_selective_scan_bwd_oracle_vs_reference_autograd(name):
    g = np.load(os.path.join(GOLD, name + ".npz"))
    c = bwd_cases.scan_inputs(name)
    n = lambda t: None if t is None else t.numpy()
    out = zo.selective_scan(n(c["u"]), n(c["delta"]), n(c["A"]), n(c["B"]), n(c["C"]), n(c["D"]), n(c["z"]),
                            n(c["delta_bias"]), c["softplus"], dt=np.float64)
    gr = zo.selective_scan_bwd(n(c["u"]), n(c["delta"]), n(c["A"]), n(c["B"]), n(c["C"]), n(c["D"]), n(c["z"]),
                               n(c["delta_bias"]), n(c["dout"]), c["softplus"])
    gr["out"] = out
    L = c["u"].shape[-1]
    for key in ("out", "du", "ddelta", "dA", "dB", "dC", "dD", "dz", "ddelta_bias"):
        if key not in g.files:
            assert gr[key] is None or key == "out"
            continue
        got = gr[key]
        if name == "bwd_scan_long" and got.shape[-1] == L:
            got = got[..., bwd_cases.LONG_KEEP]
        assert rel(got, g[key]) < 2e-5, (name, key, rel(got, g[key]))      # the reference runs in fp32


@pytest.mark.parametrize("name", ["bwd_conv_silu", "bwd_conv_plain"])
def test_conv_bwd_oracle_vs_reference_autograd(name):
    g = np.load(os.path.join(GOLD, name + ".npz"))
    act = "silu" if int(g["silu"]) else None
    assert rel(zo.causal_conv1d(g["x"], g["weight"], g["bias"], act, dt=np.float64), g["out"]) < 1e-6
    dx, dw, db = zo.causal_conv1d_bwd(g["x"], g["weight"], g["bias"], g["dout"], act)
    assert rel(dx, g["dx"]) < 1e-6 and rel(dw, g["dweight"]) < 1e-6 and rel(db, g["dbias"]) < 1e-6


@pytest.mark.parametrize("name", ["bwd_norm_rms", "bwd_norm_ln", "bwd_norm_rms_nores"])
def test_norm_bwd_oracle_vs_reference_autograd(name):
    g = np.load(os.path.join(GOLD, name + ".npz"))
    res = g["residual"] if "residual" in g.files else None
    bias = g["bias"] if "bias" in g.files else None
    dx, dw, db, dres = zo.fused_add_norm_bwd(g["x"], g["weight"], bias, res, g["dy"], g["dresidual_out"],
                                             eps=float(g["eps"]), rms=bool(int(g["rms"])))
    assert rel(dx, g["dx"]) < 1e-6 and rel(dw, g["dweight"]) < 1e-6
    if bias is not None:
        assert rel(db, g["dbias"]) < 1e-6
    if res is not None:
        assert rel(dres, g["dresidual"]) < 1e-6
