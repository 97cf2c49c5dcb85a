"""zigma_in_conv_x_proj_fwd (round 4): the x half of Mamba.in_proj + gather + conv + SiLU + x_proj in one kernel, against
  (a) a float64 evaluation of the reference's formulas on the same bf16 operands (mamba_simple.py:290-294,362-370;
      selective_scan_interface.py:307-322) with the reference's rounding points (x and u are bf16 tensors there),
  (b) the kernels it replaces (library in_proj GEMM + zigma_conv_x_proj_fwd), and
  (c) at module level: Mamba / mamba_inner_hidden against mamba_inner_tok on the materialised xz, with the call trace.
Shapes cover one tile per workgroup (every workgroup but the sequence starts takes its causal window from the pre-pass),
two tiles per workgroup (the carry tile), long sequences, and all row-table kinds."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import rel_err

pytestmark = pytest.mark.gpu
DEV = "cuda"
BF = torch.bfloat16


def N(t):
    return t.detach().double().cpu().numpy()


@pytest.fixture(scope="module", autouse=True)
def _lib_loaded():
    from zigma_amd import _lib
    _lib.lib()
    assert torch.cuda.is_available()


def _operands(B, L, E, Di, n, seed):
    g = torch.Generator(device="cpu").manual_seed(seed)
    r = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc).to(BF).to(DEV)
    return dict(h=r(B, L, E), w_in=r(2 * Di, E, sc=E ** -0.5), conv_w=r(Di, 4, sc=0.5), conv_b=r(Di, sc=0.5), w_x=r(n, Di, sc=Di ** -0.5))


def _table(kind, L):
    if kind == "none":
        return None
    if kind == "zigzag":
        from zigma_amd import scan_paths
        side = int(round(L ** 0.5))
        return torch.as_tensor(np.ascontiguousarray(scan_paths.zigzag_path(side)[5]), dtype=torch.int32, device=DEV)
    if kind == "reversed":
        return torch.arange(L - 1, -1, -1, dtype=torch.int32, device=DEV)
    return torch.randperm(L, generator=torch.Generator().manual_seed(7)).to(torch.int32).to(DEV)


def _reference_f64(op, perm, rows):
    """float64 on the same bf16 operands, rounded where the reference's bf16 tensors round (x, u); only `rows` samples"""
    h, w_in, cw, cb, w_x = (N(op[k]) for k in ("h", "w_in", "conv_w", "conv_b", "w_x"))
    Di = cw.shape[0]
    hb = h[rows]
    x = torch.from_numpy(hb @ w_in[:Di].T).to(BF).double().numpy()                      # (r, L, Di), a bf16 tensor in the reference
    if perm is not None:
        x = x[:, perm.cpu().numpy().astype(np.int64)]
    xp = np.concatenate([np.zeros((x.shape[0], 3, Di)), x], axis=1)
    L = x.shape[1]
    a = cb[None, None, :] + sum(xp[:, w:w + L] * cw[None, None, :, w] for w in range(4))
    u = a / (1.0 + np.exp(-a))
    return u


SHAPES = [  # (B, L, table)
    (32, 1024, "zigzag"),       # 256 workgroups x 1 tile: 7 of 8 take their window from the pre-pass
    (64, 1024, "random"),       # 2 tiles per workgroup: the carry tile
    (64, 1024, "none"),
    (2, 16384, "zigzag"),       # long sequences, 1 tile per workgroup
    (4, 16384, "reversed"),     # 2 tiles per workgroup on long sequences
]


@pytest.mark.parametrize("B,L,kind", SHAPES)
def test_in_conv_x_proj_vs_float64_and_replaced_kernels(B, L, kind):
    from zigma_amd.selective_scan_interface import conv_x_proj, in_conv_x_proj, in_conv_x_proj_eligible
    E, Di, n = 640, 1280, 72
    op = _operands(B, L, E, Di, n, seed=B + L)
    perm = _table(kind, L)
    assert in_conv_x_proj_eligible(op["h"], op["w_in"][:Di], op["conv_w"], op["conv_b"], op["w_x"], perm)
    u, xd = in_conv_x_proj(op["h"], op["w_in"][:Di], op["conv_w"], op["conv_b"], op["w_x"], perm)
    torch.cuda.synchronize()
    assert u.shape == (B, L, Di) and xd.shape == (B, L, n) and torch.isfinite(u.float()).all() and torch.isfinite(xd.float()).all()
    # (a) float64 on sampled batch rows (first, last: sequence starts, segment starts and tile seams are in every row)
    rows = sorted({0, B - 1})
    u_ref = _reference_f64(op, perm, rows)
    e_u = rel_err(N(u[rows]), u_ref)
    # x_dbl against float64 on the kernel's OWN u (u is a bf16 tensor in the reference as well)
    xd_ref = N(u[rows]) @ N(op["w_x"]).T
    e_x = rel_err(N(xd[rows]), xd_ref)
    print(f"B={B} L={L} {kind}: u vs float64 {e_u:.2e}, x_dbl vs float64 on own u {e_x:.2e}")
    assert e_u < 4e-3 and e_x < 3e-3, (e_u, e_x)                         # bf16 rounding of x (before the conv) and of the outputs
    # elementwise: a bf16 output is off by at most a few ulps where x sat on a rounding boundary
    du = np.abs(N(u[rows]) - u_ref)
    assert (du <= 0.04 * np.abs(u_ref) + 0.02).all(), float(du.max())
    # (b) the kernels it replaces: the same MFMA shape and summation order -> the same bf16 tensors
    xz = F.linear(op["h"], op["w_in"])
    u2, xd2 = conv_x_proj(xz[:, :, :Di], op["conv_w"], op["conv_b"], op["w_x"], perm)
    frac = float((u != u2).float().mean())
    e2 = rel_err(N(u), N(u2))
    e3 = rel_err(N(xd), N(xd2))
    print(f"   vs library in_proj + conv_x_proj: u differs in {frac:.2e} of the elements ({e2:.2e}), x_dbl {e3:.2e}")
    assert frac < 2e-2 and e2 < 2e-3 and e3 < 2e-3, (frac, e2, e3)


def test_in_conv_x_proj_limits():
    from zigma_amd.selective_scan_interface import in_conv_x_proj, in_conv_x_proj_eligible
    op = _operands(32, 1024, 768, 1536, 80, seed=3)
    assert not in_conv_x_proj_eligible(op["h"], op["w_in"][:1536], op["conv_w"], op["conv_b"], op["w_x"], None)      # d_model 768
    with pytest.raises(RuntimeError):
        in_conv_x_proj(op["h"], op["w_in"][:1536], op["conv_w"], op["conv_b"], op["w_x"], None)
    op = _operands(8, 1024, 640, 1280, 72, seed=4)
    assert not in_conv_x_proj_eligible(op["h"], op["w_in"][:1280], op["conv_w"], op["conv_b"], op["w_x"], None)      # too few positions
    op = _operands(32, 1024, 640, 1280, 96, seed=5)
    assert not in_conv_x_proj_eligible(op["h"], op["w_in"][:1280], op["conv_w"], op["conv_b"], op["w_x"], None)      # n > 80
    with pytest.raises(RuntimeError):
        in_conv_x_proj(op["h"], op["w_in"][:1280], op["conv_w"], op["conv_b"], op["w_x"], None)
    with pytest.raises(RuntimeError):       # no CPU fallback
        in_conv_x_proj(op["h"].cpu(), op["w_in"][:1280].cpu(), op["conv_w"].cpu(), op["conv_b"].cpu(), op["w_x"][:72].cpu(), None)


@pytest.mark.parametrize("kind", ["zigzag", "none"])
def test_mamba_layer_hidden_path_vs_xz_path(kind, monkeypatch):
    """Mamba.forward through mamba_inner_hidden (no xz) against the same layer through in_proj + mamba_inner_tok, bf16, with the
    call trace: one zigma_in_conv_x_proj_fwd, one z projection (n = d_inner) on the own kernel, no conv_x_proj."""
    import zigma_amd.selective_scan_interface as ssi
    from zigma_amd import _lib, scan_paths
    from zigma_amd.mamba_simple import Mamba
    B, L, E = 32, 1024, 640
    torch.manual_seed(11)
    kw = dict(scan_type="v1") if kind == "none" else dict(scan_type="zigzagN8", layer_idx=3,
                                                          zigzag_paths=[torch.as_tensor(np.ascontiguousarray(t)) for t in scan_paths.zigzag_path(32)],
                                                          zigzag_paths_reverse=[torch.as_tensor(np.ascontiguousarray(scan_paths.reverse_permut_np(t)))
                                                                                for t in scan_paths.zigzag_path(32)])
    m = Mamba(E, device=DEV, dtype=BF, **kw).eval()
    x = torch.randn(B, L, E, device=DEV).to(BF)
    trace = []
    monkeypatch.setattr(_lib, "TRACE", trace)
    with torch.no_grad():
        y_new = m(x)
    monkeypatch.setattr(_lib, "TRACE", None)
    names = [fn for fn, _, _ in trace]
    assert names.count("zigma_in_conv_x_proj_fwd") == 1 and "zigma_conv_x_proj_fwd" not in names, names
    z_calls = [P for fn, _, P in trace if fn == "zigma_linear_fwd" and P.n == m.d_inner and P.k == E]
    assert len(z_calls) == 1, names
    monkeypatch.setattr(ssi, "USE_IN_CONV_X_PROJ", False)
    trace2 = []
    monkeypatch.setattr(_lib, "TRACE", trace2)
    with torch.no_grad():
        y_old = m(x)
    monkeypatch.setattr(_lib, "TRACE", None)
    assert "zigma_in_conv_x_proj_fwd" not in [fn for fn, _, _ in trace2]
    e = rel_err(N(y_new), N(y_old))
    print(f"Mamba layer [{kind}] hidden path vs xz path: {e:.2e}")
    assert e < 2e-3, e
