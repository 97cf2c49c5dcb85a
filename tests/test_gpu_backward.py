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


# ---------------------------------------------------------------------------------------------------
# conv / norm backward kernels
# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", ["bwd_conv_silu", "bwd_conv_plain"])
def test_conv_bwd_vs_reference_autograd(name):
    from zigma_amd import _lib
    from zigma_amd.causal_conv1d_interface import conv_bwd_tok
    g = load_golden(name + ".npz")
    x, dout = tok(T(g["x"])), tok(T(g["dout"]))
    dx, dw, db = conv_bwd_tok(x, T(g["weight"]), T(g["bias"]), dout, bool(int(g["silu"])))
    assert _lib.last_kernel() == "conv_bwd_tok"
    assert rel_err(N(dx.transpose(1, 2)), g["dx"]) < 2e-5
    assert rel_err(N(dw), g["dweight"]) < 2e-5 and rel_err(N(db), g["dbias"]) < 2e-5


def test_conv_bwd_with_row_table_and_bf16():
    """gathered input / scattered dx (permutation table), ragged length, bf16 operands, vs the float64 oracle."""
    from zigma_amd.causal_conv1d_interface import conv_bwd_tok
    rng = np.random.default_rng(3)
    Bsz, L, Dm = 2, 150, 256
    x = zo.bf16_round(rng.standard_normal((Bsz, L, Dm)).astype(np.float32))
    dout = zo.bf16_round(rng.standard_normal((Bsz, L, Dm)).astype(np.float32))
    w = zo.bf16_round((rng.standard_normal((Dm, 4)) * 0.5).astype(np.float32))
    b = zo.bf16_round((rng.standard_normal(Dm) * 0.2).astype(np.float32))
    perm = rng.permutation(L).astype(np.int32)
    dx, dw, db = conv_bwd_tok(T(x, torch.bfloat16), T(w, torch.bfloat16), T(b, torch.bfloat16), T(dout, torch.bfloat16), True,
                              torch.from_numpy(perm).to(DEV))
    rdx, rdw, rdb = zo.causal_conv1d_bwd(x[:, perm].transpose(0, 2, 1), w, b, dout.transpose(0, 2, 1), "silu")
    ref_dx = np.empty_like(x)
    ref_dx[:, perm] = rdx.transpose(0, 2, 1)          # dx[row[k]] = dx'[k]
    assert rel_err(N(dx), ref_dx) < 5e-3              # bf16 output rounding
    assert rel_err(N(dw), rdw) < 1e-4 and rel_err(N(db), rdb) < 1e-4


@pytest.mark.parametrize("name", ["bwd_norm_rms", "bwd_norm_ln", "bwd_norm_rms_nores"])
def test_norm_bwd_vs_reference_autograd(name):
    """LayerNormFn (HIP forward + backward) under torch autograd against the reference's pure-torch norm."""
    from zigma_amd.layernorm import layer_norm_fn
    g = load_golden(name + ".npz")
    x = T(g["x"]).requires_grad_(True)
    res = T(g["residual"]).requires_grad_(True) if "residual" in g else None
    w = T(g["weight"]).requires_grad_(True)
    b = T(g["bias"]).requires_grad_(True) if "bias" in g else None
    y, res_out = layer_norm_fn(x, w, b, residual=res, eps=float(g["eps"]), prenorm=True, is_rms_norm=bool(int(g["rms"])))
    assert rel_err(N(y), g["y"]) < 2e-6
    torch.autograd.backward([y, res_out], [T(g["dy"]), T(g["dresidual_out"])])
    assert rel_err(N(x.grad), g["dx"]) < 2e-5 and rel_err(N(w.grad), g["dweight"]) < 2e-5
    if b is not None:
        assert rel_err(N(b.grad), g["dbias"]) < 2e-5
    if res is not None:
        assert rel_err(N(res.grad), g["dresidual"]) < 2e-5


# ---------------------------------------------------------------------------------------------------
# autograd through the Mamba inner and the whole model
# ---------------------------------------------------------------------------------------------------
def test_mamba_inner_tok_grads_vs_reference_autograd():
    from zigma_amd.selective_scan_interface import mamba_inner_tok
    import torch.nn.functional as F
    g = load_golden("bwd_mamba_inner.npz")
    leaf = lambda k: T(g[k]).requires_grad_(True)
    xz_cf = leaf("xz")                                   # (B, 2Di, L) as the reference lays it out
    cw, cb, xw, dw, ow, ob, A, D, dbias = (leaf(k) for k in ("conv_w", "conv_b", "x_proj_w", "dt_proj_w", "out_proj_w",
                                                               "out_proj_b", "A", "D", "delta_bias"))
    y = mamba_inner_tok(xz_cf.transpose(1, 2).contiguous(), cw, cb, xw, dw, A, D, dbias, delta_softplus=True)
    out = F.linear(y, ow, ob)
    assert rel_err(N(out), g["out"]) < 2e-5
    out.backward(T(g["dout"]))
    for k, t in (("xz", xz_cf), ("conv_w", cw), ("conv_b", cb), ("x_proj_w", xw), ("dt_proj_w", dw), ("out_proj_w", ow),
                 ("out_proj_b", ob), ("A", A), ("D", D), ("delta_bias", dbias)):
        assert rel_err(N(t.grad), g["d_" + k]) < 1e-4, (k, rel_err(N(t.grad), g["d_" + k]))


@pytest.mark.parametrize("name", ["zigma_text_zigzag2", "zigma_uncond_zigzag8"])
def test_model_parameter_gradients_vs_reference_autograd(name):
    """loss = sum(out * w): every parameter gradient and the input gradient of the whole model (zigzag gather/scatter,
    adaLN, cross-attention, fp32 residual stream) against autograd through the unmodified reference."""
    import ast
    from zigma_amd.model_zigma import ZigMa
    g, gg = load_golden(name + ".npz"), load_golden("bwd_model_" + name + ".npz")
    cfg = ast.literal_eval(str(g["cfg"]))
    m = ZigMa(device=DEV, dtype=torch.float32, **cfg).eval()
    m.load_state_dict({k[3:]: torch.from_numpy(v) for k, v in g.items() if k.startswith("sd.")}, strict=True)
    x = T(g["x"]).requires_grad_(True)
    y = T(g["y"]) if "y" in g else None
    out = m(x, T(g["t"]), y)
    assert rel_err(N(out), g["out"]) < 1e-4
    (out * T(gg["wgt"])).sum().backward()
    assert rel_err(N(x.grad), gg["gx"]) < 5e-4
    worst = 0.0
    for k, p in m.named_parameters():
        if "g." + k not in gg:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, k
            continue
        e = rel_err(N(p.grad), gg["g." + k])
        worst = max(worst, e)
        assert e < 2e-3, (k, e)
    assert worst < 2e-3


def test_training_step_bf16_runs_and_decreases_loss():
    """A few AdamW steps of the flow-matching loss on the bf16 model through the HIP forward + backward kernels."""
    from zigma_amd.model_zigma import ZigMa
    from zigma_amd.transport import create_transport
    torch.manual_seed(0)
    m = ZigMa(in_channels=4, embed_dim=128, depth=2, img_dim=8, patch_size=1, scan_type="zigzagN8", use_pe=2, device=DEV,
              dtype=torch.bfloat16).train()
    opt = torch.optim.AdamW(m.parameters(), lr=2e-3)
    tr = create_transport()
    x1 = torch.randn(8, 4, 8, 8, device=DEV)
    losses = []
    for _ in range(12):
        opt.zero_grad(set_to_none=True)
        loss = tr.training_losses(m, x1)["loss"].mean()
        loss.backward()
        opt.step()
        losses.append(float(loss.detach()))
    assert all(np.isfinite(losses)) and losses[-1] < losses[0], losses


def test_extension_shim_backward_entry_points():
    """selective_scan_cuda.bwd / causal_conv1d_cuda.causal_conv1d_bwd with the reference's (B, D, L) operands and return
    conventions (what the reference's SelectiveScanFn.backward / CausalConv1dFn.backward call)."""
    from zigma_amd import extension_shims
    ss, cc = extension_shims.install()
    g = load_golden("bwd_scan_full.npz")
    c = bwd_cases.scan_inputs("bwd_scan_full")
    dev = lambda k: None if c[k] is None else c[k].to(DEV)
    u, delta, A, D, z, db, dout = (dev(k) for k in ("u", "delta", "A", "D", "z", "delta_bias", "dout"))
    Bm, Cm = dev("B").unsqueeze(1), dev("C").unsqueeze(1)
    out, x, out_z = ss.fwd(u, delta, A, Bm, Cm, D, z, db, True)
    dz_buf = torch.empty_like(z)
    du, ddelta, dA, dB, dC, dD, dbias, dz, oz = ss.bwd(u, delta, A, Bm, Cm, D, z, db, dout, x, out, dz_buf, True, True)
    assert dz.data_ptr() == dz_buf.data_ptr() and dB.shape == Bm.shape and dB.dtype == torch.float32
    assert torch.allclose(oz, out_z, rtol=1e-5, atol=1e-6)
    for got, key in ((du, "du"), (ddelta, "ddelta"), (dA, "dA"), (dB[:, 0], "dB"), (dC[:, 0], "dC"), (dD, "dD"),
                     (dbias, "ddelta_bias"), (dz, "dz")):
        assert rel_err(N(got), g[key]) < 5e-5, key
    gc = load_golden("bwd_conv_silu.npz")
    dx, dw, dbb = cc.causal_conv1d_bwd(T(gc["x"]), T(gc["weight"]), T(gc["bias"]), T(gc["dout"]), None, True)
    assert dx.shape == gc["dx"].shape and rel_err(N(dx), gc["dx"]) < 2e-5 and rel_err(N(dw), gc["dweight"]) < 2e-5
    with pytest.raises(NotImplementedError):
        cc.causal_conv1d_update(None)


def test_mamba_inner_tok_row_tables_match_explicit_gather_scatter():
    """Row tables fused into the forward AND backward kernels (gather table perm, write-back table out_rows that is NOT its
    inverse, as in the video temporal layers) vs the same op on explicitly gathered rows + an explicit scatter."""
    from zigma_amd.selective_scan_interface import mamba_inner_tok
    g = torch.Generator(device="cpu").manual_seed(9)
    Bsz, L, Di, R, Nst = 2, 48, 128, 8, 16
    mk = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc).to(DEV).requires_grad_(True)
    xz, cw, cb = mk(Bsz, L, 2 * Di), mk(Di, 1, 4, sc=0.5), mk(Di, sc=0.1)
    xw, dw = mk(R + 2 * Nst, Di, sc=Di ** -0.5), mk(Di, R, sc=R ** -0.5)
    A = (-torch.exp(torch.randn(Di, Nst, generator=g) * 0.5)).to(DEV).requires_grad_(True)
    D, db = mk(Di), (torch.rand(Di, generator=g) * 0.5).to(DEV).requires_grad_(True)
    perm = torch.randperm(L, generator=g).to(DEV, torch.int32)
    out_rows = torch.randperm(L, generator=g).to(DEV, torch.int32)
    wgt = torch.randn(Bsz, L, Di, generator=g).to(DEV)
    leaves = (xz, cw, cb, xw, dw, A, D, db)

    y1 = mamba_inner_tok(xz, cw, cb, xw, dw, A, D, db, perm=perm, out_rows=out_rows)
    (y1 * wgt).sum().backward()
    g1 = [t.grad.clone() for t in leaves]
    for t in leaves:
        t.grad = None
    ys = mamba_inner_tok(xz.index_select(1, perm.long()), cw, cb, xw, dw, A, D, db)      # scan order in, scan order out
    y2 = torch.zeros_like(ys).index_copy(1, out_rows.long(), ys)                          # y_tok[out_rows[k]] = y'[k]
    (y2 * wgt).sum().backward()
    assert torch.allclose(y1, y2, rtol=1e-5, atol=1e-6)
    for a, t in zip(g1, leaves):
        assert rel_err(N(a), N(t.grad)) < 1e-5


def test_mamba_inner_tok_train_with_the_one_pass_conv_x_proj(monkeypatch):
    """The differentiable Mamba inner at a size where its forward takes the one-pass conv + x_proj kernel (bf16, 16 384 positions)
    vs the same op with the two separate kernels: same output and gradients up to bf16 roundings (u differs by single ulps in a
    few elements per million; everything downstream sees that)."""
    import zigma_amd.selective_scan_interface as ssi
    from zigma_amd import _lib
    g = torch.Generator(device="cpu").manual_seed(11)
    Bsz, L, Di, R, Nst = 16, 1024, 128, 8, 16
    bf = torch.bfloat16
    mk = lambda *s, sc=1.0, dt=bf: (torch.randn(*s, generator=g) * sc).to(DEV, dt).requires_grad_(True)
    xz, cw, cb = mk(Bsz, L, 2 * Di), mk(Di, 1, 4, sc=0.5), mk(Di, sc=0.1)
    xw, dw = mk(R + 2 * Nst, Di, sc=Di ** -0.5), mk(Di, R, sc=R ** -0.5)
    A = (-torch.exp(torch.randn(Di, Nst, generator=g) * 0.5)).to(DEV).requires_grad_(True)
    D, db = mk(Di, dt=torch.float32), (torch.rand(Di, generator=g) * 0.5).to(DEV).requires_grad_(True)
    perm = torch.randperm(L, generator=g).to(DEV, torch.int32)
    wgt = torch.randn(Bsz, L, Di, generator=g).to(DEV, bf)
    leaves = (xz, cw, cb, xw, dw, A, D, db)
    seen = []
    real = ssi.conv_x_proj
    monkeypatch.setattr(ssi, "conv_x_proj", lambda *a, **k: (seen.append(1), real(*a, **k))[1])
    res = []
    for fused in (True, False):
        monkeypatch.setattr(ssi, "USE_CONV_X_PROJ", fused)
        for t in leaves:
            t.grad = None
        y = ssi.mamba_inner_tok(xz, cw, cb, xw, dw, A, D, db, perm=perm)
        (y.float() * wgt.float()).sum().backward()
        res.append((y.detach().float(), [t.grad.detach().float().clone() for t in leaves]))
    assert len(seen) == 1                                            # taken with the switch on, not with it off
    (y1, g1), (y2, g2) = res
    assert rel_err(N(y1), N(y2)) < 2e-3
    for a, b in zip(g1, g2):
        assert rel_err(N(a), N(b)) < 1e-2


@pytest.mark.parametrize("Bsz,L,E", [(2, 64, 128), (3, 192, 640), (16, 1024, 640), (2, 128, 768)])
def test_glue_backward_kernel_vs_float64(Bsz, L, E):
    """zigma_scale_reduce_bwd (backward of modulate / gated add in one pass) vs the autograd formulas in float64 on the same bf16
    operands, including strided operands (a column slice of a wider buffer, a chunk of the adaLN rows)."""
    from zigma_amd import _lib
    from zigma_amd.layernorm import glue_bwd_eligible, scale_reduce_bwd
    g = torch.Generator(device="cpu").manual_seed(L + E)
    bf = torch.bfloat16
    dy = torch.randn(Bsz, L, E, generator=g).to(DEV, bf)
    a = torch.randn(Bsz, L, E + 64, generator=g).to(DEV, bf)[:, :, 64:]
    s = torch.randn(Bsz, 3 * E, generator=g).to(DEV, bf)[:, E:2 * E]
    assert glue_bwd_eligible(dy, a, s)
    out, r1, r2 = scale_reduce_bwd(dy, a, s, s_add=1.0, want_sum=True)
    assert _lib.last_kernel() == "scale_reduce_bwd"
    d64, a64, s64 = dy.double(), a.double(), s.double()
    assert rel_err(N(out), N(d64 * (1 + s64).unsqueeze(1))) < 3e-3
    assert rel_err(N(r1), N((d64 * a64).sum(1))) < 3e-3 and rel_err(N(r2), N(d64.sum(1))) < 3e-3
    out2, r1b, none = scale_reduce_bwd(dy, a, s)
    assert none is None and rel_err(N(out2), N(d64 * s64.unsqueeze(1))) < 3e-3 and torch.equal(r1b, r1)
    # through the autograd Functions of the block, against plain autograd of the same expressions
    from zigma_amd.model_zigma import _GatedAddFn, _ModulateFn
    x = torch.randn(Bsz, L, E, generator=g).to(DEV, bf).requires_grad_(True)
    sh = torch.randn(Bsz, E, generator=g).to(DEV, bf).requires_grad_(True)
    sc = torch.randn(Bsz, E, generator=g).to(DEV, bf).requires_grad_(True)
    w = torch.randn(Bsz, L, E, generator=g).to(DEV, bf)
    (_ModulateFn.apply(x, sh, sc).float() * w.float()).sum().backward()
    got = [t.grad.clone() for t in (x, sh, sc)]
    for t in (x, sh, sc):
        t.grad = None
    ((x.double() * (1 + sc.double()).unsqueeze(1) + sh.double().unsqueeze(1)) * w.double()).sum().backward()
    for a_, t in zip(got, (x, sh, sc)):
        assert rel_err(N(a_), N(t.grad)) < 4e-3
    for t in (x, sh, sc):
        t.grad = None
    (_GatedAddFn.apply(w, sh, x).float() * dy.float()).sum().backward()
    got = [t.grad.clone() for t in (sh, x)]
    for t in (x, sh):
        t.grad = None
    ((w.double() + sh.double().unsqueeze(1) * x.double()) * dy.double()).sum().backward()
    assert rel_err(N(got[0]), N(sh.grad)) < 4e-3 and rel_err(N(got[1]), N(x.grad)) < 4e-3


def test_likelihood_sampler_runs_through_hip_backward():
    """Sampler.sample_ode_likelihood needs a vjp through the denoiser at every function evaluation: here it goes through
    LayerNormFn / MambaInnerTokFn (HIP forward + backward).  Checked against a finite-difference divergence probe."""
    from zigma_amd.transport import Sampler, create_transport
    m, g, cfg, y = _tiny_model()
    fn = Sampler(create_transport()).sample_ode_likelihood(sampling_method="euler", num_steps=4)
    torch.manual_seed(0)
    x = torch.randn(2, 4, 8, 8, device=DEV)
    logp, z = fn(x, m.forward)
    assert logp.shape == (2,) and z.shape == x.shape and torch.isfinite(logp).all() and torch.isfinite(z).all()
    # the vjp itself: eps . J eps from autograd vs a central finite difference of the model along eps
    t = torch.full((2,), 0.4, device=DEV)
    eps = (torch.randint(2, x.shape, device=DEV).float() * 2 - 1)
    xg = x.clone().requires_grad_(True)
    out = m(xg, t)
    vjp = torch.autograd.grad((out * eps).sum(), xg)[0]
    quad = (vjp * eps).flatten(1).sum(1)
    h = 1e-2
    with torch.no_grad():
        fd = (((m(x + h * eps, t) - m(x - h * eps, t)) / (2 * h)) * eps).flatten(1).sum(1)
    assert torch.allclose(quad, fd, rtol=5e-2, atol=5e-2), (quad, fd)


def _tiny_model():
    import ast
    from zigma_amd.model_zigma import ZigMa
    g = load_golden("zigma_uncond_zigzag8.npz")
    cfg = ast.literal_eval(str(g["cfg"]))
    m = ZigMa(device=DEV, dtype=torch.float32, **cfg).eval()
    m.load_state_dict({k[3:]: torch.from_numpy(v) for k, v in g.items() if k.startswith("sd.")}, strict=True)
    return m, g, cfg, None


@pytest.mark.parametrize("L", [100, 4096 + 32])
def test_scan_bwd_with_forward_written_checkpoints_is_identical(L):
    """checkpoints written by the forward kernel (single pass and the sequence-split mode) vs the backward's own phase 1."""
    from zigma_amd import _lib
    from zigma_amd.selective_scan_interface import scan_bwd_tok, scan_raw
    g = torch.Generator(device="cpu").manual_seed(L)
    Bsz, Dm, Nst = 2, 128, 16
    r = lambda *s: torch.randn(*s, generator=g).to(DEV)
    u, delta, z, dout = r(Bsz, L, Dm), (0.5 * torch.rand(Bsz, L, Dm, generator=g)).to(DEV), r(Bsz, L, Dm), r(Bsz, L, Dm)
    A = (-0.5 * torch.rand(Dm, Nst, generator=g) - 0.05).to(DEV)
    Bm, Cm, D, db = r(Bsz, L, Nst), r(Bsz, L, Nst), r(Dm), torch.rand(Dm, generator=g).to(DEV)
    out, oz = torch.empty_like(u), torch.empty_like(u)
    ck = torch.full((Bsz, Dm // 64, (L + 15) // 16, Nst, 64), float("nan"), device=DEV)
    xc = torch.empty(Bsz, Dm, (L + 2047) // 2048, 2 * Nst, device=DEV) if L > 4096 else None     # lets the kernel split
    scan_raw(u.transpose(1, 2), delta.transpose(1, 2), A, Bm.transpose(1, 2).unsqueeze(1), Cm.transpose(1, 2).unsqueeze(1), D,
             z.transpose(1, 2), db, True, out=out.transpose(1, 2), out_z=oz.transpose(1, 2), checkpoints=ck, x=xc)
    assert _lib.last_kernel() == "scan_tok_n16" and torch.isfinite(ck).all()
    a = scan_bwd_tok(u, delta, A, Bm, Cm, D, z, db, dout, out, True, checkpoints=ck)
    b = scan_bwd_tok(u, delta, A, Bm, Cm, D, z, db, dout, out, True)
    for x, y in zip(a, b):
        assert torch.equal(x, y) if L <= 4096 else rel_err(N(x), N(y)) < 1e-5      # split mode: carries combine in another order


def test_model_gradients_bf16_close_to_fp32_reference():
    """bf16 parameters / activations through the HIP backward: gradients stay within bf16 resolution of the reference's
    fp32 autograd gradients (cosine similarity per parameter tensor, norm-wise error of the input gradient)."""
    import ast
    from zigma_amd.model_zigma import ZigMa
    name = "zigma_uncond_zigzag8"
    g, gg = load_golden(name + ".npz"), load_golden("bwd_model_" + name + ".npz")
    cfg = ast.literal_eval(str(g["cfg"]))
    m = ZigMa(device=DEV, dtype=torch.bfloat16, **cfg).eval()
    m.load_state_dict({k[3:]: torch.from_numpy(v) for k, v in g.items() if k.startswith("sd.")}, strict=True)
    x = T(g["x"]).requires_grad_(True)
    out = m(x, T(g["t"]), None)
    (out * T(gg["wgt"])).sum().backward()
    assert rel_err(N(x.grad), gg["gx"]) < 8e-2
    worst = 1.0
    for k, p in m.named_parameters():
        if "g." + k not in gg or p.grad is None:
            continue
        a, b = N(p.grad).ravel().astype(np.float64), gg["g." + k].ravel().astype(np.float64)
        if np.linalg.norm(b) < 1e-6:
            continue
        worst = min(worst, float(a @ b / (np.linalg.norm(a) * np.linalg.norm(b) + 1e-30)))
    assert worst > 0.95, worst          # measured 0.97 (a small, cancellation-heavy tensor); fp32 gradients match to 2e-3


@pytest.mark.parametrize("Bsz,L,H,NC", [(2, 256, 8, 77), (3, 100, 4, 5)])
def test_cross_attn_train_gradients_and_no_fused_sdpa(Bsz, L, H, NC, monkeypatch):
    """The differentiable cross-attention of the training path (zigma_amd.attention.CrossAttnFn: HIP forward and backward kernels)
    against autograd through a float64 evaluation of the reference's
    scaled_dot_product_attention (model_zigma.py:113-127) on the same bf16 operands; and torch's fused SDPA (AOT-Triton on ROCm) is
    never reached — neither here nor by a whole training step of a text model."""
    import torch.nn.functional as F
    from zigma_amd.attention import cross_attn_train

    def boom(*a, **k):
        raise AssertionError("fused scaled_dot_product_attention was called")
    g = torch.Generator(device="cpu").manual_seed(L + NC)
    C = H * 64
    mk = lambda *s: torch.randn(*s, generator=g).to(DEV, torch.bfloat16)
    q, k, v, do = mk(Bsz, L, C), mk(Bsz, NC, C), mk(Bsz, NC, C), mk(Bsz, L, C)
    qd, kd, vd = (t.double().requires_grad_(True) for t in (q, k, v))
    hd = lambda t: t.view(t.shape[0], t.shape[1], H, 64).transpose(1, 2)
    ref = F.scaled_dot_product_attention(hd(qd), hd(kd), hd(vd)).transpose(1, 2).reshape(Bsz, L, C)
    g_ref = torch.autograd.grad(ref, (qd, kd, vd), do.double())
    monkeypatch.setattr(F, "scaled_dot_product_attention", boom)
    q1, k1, v1 = (t.clone().requires_grad_(True) for t in (q, k, v))
    out = cross_attn_train(q1, k1, v1, H)
    out.backward(do)
    assert rel_err(N(out), N(ref)) < 8e-3
    for name, got, want in zip("qkv", (q1.grad, k1.grad, v1.grad), g_ref):
        assert rel_err(N(got), N(want)) < 1.5e-2, name           # bf16 probabilities / dS, fp32 accumulation
    # a whole training step of a text model never reaches the fused SDPA
    from zigma_amd.model_zigma import ZigMa
    m = ZigMa(in_channels=4, embed_dim=64, depth=2, img_dim=8, patch_size=1, has_text=True, d_context=32, n_context_token=7,
              scan_type="zigzagN2", use_pe=2, device=DEV, dtype=torch.bfloat16).train()
    with torch.no_grad():
        for blk in m.blocks:
            blk.adaLN_modulation[-1].bias.normal_(std=0.3)
    x = torch.randn(2, 4, 8, 8, device=DEV, dtype=torch.bfloat16)
    loss = m(x, torch.rand(2, device=DEV), torch.rand(2, 7, 32, device=DEV, dtype=torch.bfloat16)).float().square().mean()
    loss.backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in m.blocks[0].msa.parameters())


@pytest.mark.parametrize("Bsz,L,H,NC", [(2, 1024, 8, 77), (1, 512, 2, 80), (3, 100, 4, 5), (2, 17, 1, 1), (1, 640, 3, 128), (2, 300, 2, 97)])
def test_cross_attn_bwd_kernel_vs_float64(Bsz, L, H, NC):
    """zigma_cross_attn_bwd against float64 autograd through softmax(scale q k^T) v (reference model_zigma.py:113-127) on the same bf16
    operands: ragged token tiles, one / several chunks of tokens, both key-block instantiations, K / V as column slices of one batched
    projection (the layout the block produces); and against the GEMM + ATen composition of the same formulas."""
    from zigma_amd import _lib
    from zigma_amd.attention import cross_attn_bwd, cross_attn_bwd_math
    g = torch.Generator(device="cpu").manual_seed(7 * L + NC)
    C = H * 64
    mk = lambda *s: torch.randn(*s, generator=g).to(DEV, torch.bfloat16)
    q, do, kv = mk(Bsz, L, C), mk(Bsz, L, C), mk(Bsz, NC, 2 * C) * 1.5
    k, v = kv[..., :C], kv[..., C:]
    _lib.TRACE = []
    dq, dk, dv = cross_attn_bwd(q, k, v, do, H)
    trace, _lib.TRACE = _lib.TRACE, None
    assert [t[1] for t in trace] == ["cross_attn_bwd_mfma"]
    qd, kd, vd = (t.double().requires_grad_(True) for t in (q, k, v))
    hd = lambda t: t.reshape(t.shape[0], t.shape[1], H, 64).transpose(1, 2)
    p = torch.softmax(hd(qd) @ hd(kd).transpose(-1, -2) * 64 ** -0.5, -1)
    ref = (p @ hd(vd)).transpose(1, 2).reshape(Bsz, L, C)
    want = torch.autograd.grad(ref, (qd, kd, vd), do.double())
    comp = cross_attn_bwd_math(q, k, v, do, H, 64 ** -0.5)
    for name, got, w, c in zip("qkv", (dq, dk, dv), want, comp):
        assert torch.isfinite(got).all(), name
        assert rel_err(N(got), N(w)) < 1e-2, (name, rel_err(N(got), N(w)))      # P / dS rounded to bf16 as MFMA operands, fp32 sums
        assert rel_err(N(got), N(c)) < 1e-2, (name, rel_err(N(got), N(c)))
    # deterministic: fixed-order sums, no atomics
    dq2, dk2, dv2 = cross_attn_bwd(q, k, v, do, H)
    assert torch.equal(dq, dq2) and torch.equal(dk, dk2) and torch.equal(dv, dv2)


@pytest.mark.parametrize("with_forward_checkpoints", [False, True])
def test_scan_bwd_reset_period_vs_oracle_on_separate_sequences(with_forward_checkpoints):
    """reset_period in the backward (the differentiable video temporal layers, reference mamba_simple.py:420-440 trained without the
    transposing copies): a batch row that concatenates R-step sequences must give, step for step, the gradients the float64 oracle
    (selective_scan_bwd_kernel.cuh:161-329 restated) gives for the same sequences as separate batch rows — with the checkpoints the
    forward kernel wrote and with the backward's own first phase."""
    from zigma_amd.selective_scan_interface import scan_bwd_tok, scan_raw
    rng = np.random.default_rng(11)
    Bsz, S, R, Dm, Nst = 3, 4, 16, 128, 16          # 3 rows of 4 sequences of 16 steps
    L = S * R
    mk = lambda *s: zo.bf16_round(rng.standard_normal(s).astype(np.float32))
    c = dict(u=mk(Bsz, Dm, L), delta=zo.bf16_round((0.5 * rng.random((Bsz, Dm, L))).astype(np.float32)),
             A=(-0.5 * rng.random((Dm, Nst)) - 0.05).astype(np.float32), B=mk(Bsz, Nst, L), C=mk(Bsz, Nst, L),
             D=rng.standard_normal(Dm).astype(np.float32), z=mk(Bsz, Dm, L), delta_bias=(0.5 * rng.random(Dm)).astype(np.float32),
             dout=mk(Bsz, Dm, L))
    dt = torch.bfloat16
    u, delta, z, dout, Bm, Cm = (tok(T(c[k], dt)) for k in ("u", "delta", "z", "dout", "B", "C"))
    A, D, db = T(c["A"]), T(c["D"]), T(c["delta_bias"])
    out, oz = torch.empty_like(u), torch.empty_like(u)
    ck = torch.empty(Bsz, Dm // 64, L // 16, Nst, 64, device=DEV, dtype=torch.float32) if with_forward_checkpoints else None
    info = []
    scan_raw(u.transpose(1, 2), delta.transpose(1, 2), A, Bm.transpose(1, 2).unsqueeze(1), Cm.transpose(1, 2).unsqueeze(1), D,
             z.transpose(1, 2), db, True, out=out.transpose(1, 2), out_z=oz.transpose(1, 2), checkpoints=ck, reset_period=R, info=info)
    if with_forward_checkpoints:
        assert info[1] == 1            # the forward kernel wrote them
    du, ddelta, dA, dB, dC, dD, dz, dbias = scan_bwd_tok(u, delta, A, Bm, Cm, D, z, db, dout, out, True, checkpoints=ck, reset_period=R)
    # the same sequences as separate rows: (B, D, S*R) -> (B*S, D, R)
    sep = lambda a: a.reshape(a.shape[0], a.shape[1], S, R).transpose(0, 2, 1, 3).reshape(a.shape[0] * S, a.shape[1], R)
    ref = zo.selective_scan_bwd(sep(c["u"]), sep(c["delta"]), c["A"], sep(c["B"]), sep(c["C"]), c["D"], sep(c["z"]), c["delta_bias"],
                                sep(c["dout"]), True)
    cat = lambda a: a.reshape(Bsz, S, a.shape[1], R).transpose(0, 2, 1, 3).reshape(Bsz, a.shape[1], L)
    for got, key, tol in ((du.transpose(1, 2), "du", 1e-2), (ddelta.transpose(1, 2), "ddelta", 1e-2), (dz.transpose(1, 2), "dz", 1e-2),
                          (dB.transpose(1, 2), "dB", 2e-3), (dC.transpose(1, 2), "dC", 2e-3)):
        assert rel_err(N(got), cat(ref[key])) < tol, key
    for got, key in ((dA, "dA"), (dD, "dD"), (dbias, "ddelta_bias")):
        assert rel_err(N(got), ref[key]) < 2e-3, key


def test_conv_bwd_reset_period_vs_oracle_on_separate_sequences():
    """reset_period in the conv backward: the window (and the gradient) never crosses a sequence boundary; with a row table."""
    from zigma_amd.causal_conv1d_interface import conv_bwd_tok
    rng = np.random.default_rng(5)
    Bsz, S, R, Dm = 2, 5, 16, 256
    L = S * R
    x = zo.bf16_round(rng.standard_normal((Bsz, L, Dm)).astype(np.float32))
    dout = zo.bf16_round(rng.standard_normal((Bsz, L, Dm)).astype(np.float32))
    w = zo.bf16_round((rng.standard_normal((Dm, 4)) * 0.5).astype(np.float32))
    b = zo.bf16_round((rng.standard_normal(Dm) * 0.2).astype(np.float32))
    perm = np.concatenate([s * R + rng.permutation(R) for s in range(S)]).astype(np.int32)       # reordering inside every sequence
    dx, dw, db = conv_bwd_tok(T(x, torch.bfloat16), T(w, torch.bfloat16), T(b, torch.bfloat16), T(dout, torch.bfloat16), True,
                              torch.from_numpy(perm).to(DEV), reset_period=R)
    xs = x[:, perm].reshape(Bsz * S, R, Dm).transpose(0, 2, 1)
    rdx, rdw, rdb = zo.causal_conv1d_bwd(xs, w, b, dout.reshape(Bsz * S, R, Dm).transpose(0, 2, 1), "silu")
    ref_dx = np.empty_like(x)
    ref_dx[:, perm] = rdx.transpose(0, 2, 1).reshape(Bsz, L, Dm)
    assert rel_err(N(dx), ref_dx) < 5e-3
    assert rel_err(N(dw), rdw) < 1e-4 and rel_err(N(db), rdb) < 1e-4


def test_video_temporal_layers_train_without_transposing_copies():
    """A video model (zzvideo_sst-like: spatial and temporal layers, 16 frames) trained through the strided-view form of its temporal
    layers (reset_period in conv / scan forward AND backward, results and d(xz) as views) gives the same output and the same parameter /
    input gradients as the transposing-copy form of the reference (mamba_simple.py:420-440), whose kernels the tests above pin; and the
    strided form really ran (reset_period seen by the backward entry points)."""
    from zigma_amd import _lib, mamba_simple
    from zigma_amd.model_zigma import ZigMa
    torch.manual_seed(0)
    m = ZigMa(in_channels=4, embed_dim=128, depth=4, img_dim=8, patch_size=1, num_classes=5, scan_type="zzvideo_sst", video_frames=16,
              use_pe=2, device=DEV, dtype=torch.bfloat16).train()
    with torch.no_grad():
        for blk in m.blocks:
            blk.adaLN_modulation[-1].weight.normal_(std=0.05)
            blk.adaLN_modulation[-1].bias.normal_(std=0.3)
    x = torch.randn(2, 16, 4, 8, 8, device=DEV, dtype=torch.bfloat16)
    t, y = torch.rand(2, device=DEV), torch.tensor([1, 3], device=DEV)
    res = {}
    for mode in (True, False):
        mamba_simple.NO_COPY_TEMPORAL = mode
        try:
            m.zero_grad(set_to_none=True)
            xin = x.clone().requires_grad_(True)
            _lib.TRACE = []
            out = m(xin, t, y)
            (out.float() * torch.linspace(-1, 1, out.numel(), device=DEV).view(out.shape)).sum().backward()
            trace, _lib.TRACE = _lib.TRACE, None
        finally:
            mamba_simple.NO_COPY_TEMPORAL = True
        resets = [int(getattr(pb, "reset_period", 0)) for fn, _, pb in trace if fn in ("zigma_selective_scan_bwd", "zigma_causal_conv1d_bwd")]
        res[mode] = (out.detach().float(), xin.grad.float(), {n: p.grad.float() for n, p in m.named_parameters() if p.grad is not None}, resets)
    assert 16 in res[True][3] and not any(res[False][3])
    assert rel_err(N(res[True][0]), N(res[False][0])) < 2e-3
    assert rel_err(N(res[True][1]), N(res[False][1])) < 1e-2
    assert res[True][2].keys() == res[False][2].keys()
    worst = max((rel_err(N(res[True][2][n]), N(res[False][2][n])), n) for n in res[True][2] if float(res[False][2][n].abs().max()) > 0)
    assert worst[0] < 2e-2, worst


@pytest.mark.parametrize("M,K,N,bias", [(65536, 640, 2560, False), (16384, 1280, 640, False), (8192, 512, 640, True), (4096, 640, 512, False)])
def test_linear_train_fn_on_the_own_kernels(M, K, N, bias, monkeypatch):
    """wgrad.LinearTrainFn (round 4): forward product and dX on zigma_linear_fwd (weight-stationary / tiled kernels) — asserted from the call
    trace — against float64 on the same bf16 operands and against the library path (F.linear + autograd) within bf16 bounds; dW through
    the slab-wise product as before (reference: autograd's linear backward behind mamba_simple.py:290-294, model_zigma.py:104-135)."""
    import zigma_amd.wgrad as wg
    from zigma_amd import _lib
    g = torch.Generator(device="cpu").manual_seed(M + K)
    x = torch.randn(M, K, generator=g).to("cuda", torch.bfloat16).requires_grad_(True)
    w = (torch.randn(N, K, generator=g) * K ** -0.5).to("cuda", torch.bfloat16).requires_grad_(True)
    b = (torch.randn(N, generator=g) * 0.1).to("cuda", torch.bfloat16).requires_grad_(True) if bias else None
    dy = torch.randn(M, N, generator=g).to("cuda", torch.bfloat16)
    monkeypatch.setattr(wg, "OWN_TRAIN_GEMMS", True)
    trace = []
    monkeypatch.setattr(_lib, "TRACE", trace)
    y = wg.linear_train(x, w, b)
    y.backward(dy)
    monkeypatch.setattr(_lib, "TRACE", None)
    n_own = sum(1 for fn, _, _ in trace if fn == "zigma_linear_fwd")
    assert n_own == 2, [t[:2] for t in trace]                     # forward and dX
    got = (y.detach(), x.grad.clone(), w.grad.clone(), None if b is None else b.grad.clone())
    x.grad = w.grad = None
    if b is not None:
        b.grad = None
    monkeypatch.setattr(wg, "OWN_TRAIN_GEMMS", False)
    y2 = wg.linear_train(x, w, b)
    y2.backward(dy)
    rows = torch.randint(0, M, (512,), generator=g).to("cuda")
    xd, wd, dyd = x.detach().double(), w.detach().double(), dy.double()
    ref_y = xd[rows] @ wd.T + (0 if b is None else b.detach().double())
    ref_dx = dyd[rows] @ wd
    assert rel_err(got[0][rows].double().cpu().numpy(), ref_y.cpu().numpy()) < 2.5e-3
    assert rel_err(got[1][rows].double().cpu().numpy(), ref_dx.cpu().numpy()) < 2.5e-3
    assert rel_err(got[0].float().cpu().numpy(), y2.detach().float().cpu().numpy()) < 2e-3
    assert rel_err(got[1].float().cpu().numpy(), x.grad.float().cpu().numpy()) < 2e-3
    assert torch.equal(got[2], w.grad)                            # (the weight gradient does not depend on the switch)
    if b is not None:
        assert torch.equal(got[3], b.grad)
