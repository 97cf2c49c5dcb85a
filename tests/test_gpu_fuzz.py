"""Seeded random-shape sweep of the HIP forward kernels against the oracle: ragged lengths, odd batch sizes, padded row
pitches, row tables, both state sizes, the three I/O types.  Complements the hand-picked cases of test_gpu_parity.py."""
import numpy as np
import pytest
import torch

from conftest import rel_err
from oracle import zigma_oracle as zo

pytestmark = pytest.mark.gpu
DEV = "cuda"
DT = {"f32": torch.float32, "bf16": torch.bfloat16, "f16": torch.float16}


def _round(a, kind):
    if kind == "bf16":
        return zo.bf16_round(a)
    if kind == "f16":
        return a.astype(np.float16).astype(np.float32)
    return a


def _padded(a, pad, dtype):
    """device tensor whose last-but-one stride is padded (row pitch > row length), as slices of bigger buffers are"""
    t = torch.from_numpy(np.ascontiguousarray(a))
    buf = torch.zeros(*t.shape[:-1], t.shape[-1] + pad, dtype=dtype, device=DEV)
    buf[..., :t.shape[-1]] = t.to(DEV).to(dtype)
    return buf[..., :t.shape[-1]]


@pytest.mark.parametrize("seed", range(24))
def test_scan_tok_random_shapes(seed):
    from zigma_amd import _lib
    from zigma_amd.selective_scan_interface import scan_raw
    rng = np.random.default_rng(1000 + seed)
    kind = ["f32", "bf16", "f16"][seed % 3]
    Bsz, Dm, Nst = int(rng.integers(1, 4)), 64 * int(rng.integers(1, 4)), [16, 8][seed % 2]
    L = int(rng.choice([1, 3, 15, 16, 17, 31, 48, 100, 129, 257]))
    has_z, has_D, has_bias, softplus, tables = (bool(rng.integers(0, 2)) for _ in range(5))
    pad = 8 * int(rng.integers(0, 3))
    r = lambda *s: _round(rng.standard_normal(s).astype(np.float32), kind)
    u, delta = r(Bsz, L, Dm), _round((0.5 * rng.random((Bsz, L, Dm))).astype(np.float32), kind)
    z = r(Bsz, L, Dm) if has_z else None
    A = (-0.5 * rng.random((Dm, Nst)) - 0.05).astype(np.float32)
    Bm, Cm = r(Bsz, L, Nst), r(Bsz, L, Nst)
    D = rng.standard_normal(Dm).astype(np.float32) if has_D else None
    db = (0.5 * rng.random(Dm)).astype(np.float32) if has_bias else None
    perm = rng.permutation(L).astype(np.int32) if (tables and has_z) else None
    dt = DT[kind]
    T = lambda a: None if a is None else torch.from_numpy(a).to(DEV)
    ut, dtt = _padded(u, pad, dt), _padded(delta, pad, dt)
    zt = _padded(z, pad, dt) if has_z else None
    Bt, Ct = _padded(Bm, 8, dt), _padded(Cm, 8, dt)
    y = torch.empty(Bsz, L, Dm, device=DEV, dtype=dt)
    pt = None if perm is None else torch.from_numpy(perm).to(DEV)
    kw = dict(out_z=y.transpose(1, 2), want_out=False) if has_z else dict(out=y.transpose(1, 2))
    scan_raw(ut.transpose(1, 2), dtt.transpose(1, 2), T(A), Bt.transpose(1, 2).unsqueeze(1), Ct.transpose(1, 2).unsqueeze(1), T(D),
             None if zt is None else zt.transpose(1, 2), T(db), softplus, z_row_index=pt, out_row_index=pt, **kw)
    assert _lib.last_kernel().startswith("scan_tok")
    zs = None if z is None else (z if perm is None else z[:, perm])
    ref = zo.selective_scan(u.transpose(0, 2, 1), delta.transpose(0, 2, 1), A, Bm.transpose(0, 2, 1), Cm.transpose(0, 2, 1), D,
                            None if zs is None else zs.transpose(0, 2, 1), db, softplus).transpose(0, 2, 1)
    if perm is not None:
        full = np.empty_like(ref)
        full[:, perm] = ref
        ref = full
    tol = 2e-5 if kind == "f32" else (1e-2 if kind == "bf16" else 2e-3)
    got = y.float().cpu().numpy()
    assert np.isfinite(got).all() and rel_err(got, _round(ref, kind)) < tol, (kind, Bsz, Dm, Nst, L, has_z, tables)


@pytest.mark.parametrize("seed", range(12))
def test_conv_tok_random_shapes(seed):
    from zigma_amd.causal_conv1d_interface import causal_conv1d_raw
    rng = np.random.default_rng(2000 + seed)
    kind = ["f32", "bf16", "f16"][seed % 3]
    Bsz, Dm, W = int(rng.integers(1, 4)), 4 * int(rng.integers(1, 80)), int(rng.integers(2, 5))
    L = int(rng.choice([1, 2, 3, 15, 16, 17, 40, 100, 130]))
    silu, tables, has_bias = (bool(rng.integers(0, 2)) for _ in range(3))
    x = _round(rng.standard_normal((Bsz, L, Dm)).astype(np.float32), kind)
    w = _round((rng.standard_normal((Dm, W)) * 0.5).astype(np.float32), kind)
    b = _round((rng.standard_normal(Dm) * 0.2).astype(np.float32), kind) if has_bias else None
    perm = rng.permutation(L).astype(np.int32) if tables else None
    dt = DT[kind]
    xt = _padded(x, 4 * int(rng.integers(0, 3)), dt)
    out = torch.empty(Bsz, L, Dm, device=DEV, dtype=dt)
    causal_conv1d_raw(xt.transpose(1, 2), torch.from_numpy(w).to(DEV, dt), None if b is None else torch.from_numpy(b).to(DEV, dt), silu,
                      out=out.transpose(1, 2), x_row_index=None if perm is None else torch.from_numpy(perm).to(DEV))
    xs = x if perm is None else x[:, perm]
    ref = zo.causal_conv1d(xs.transpose(0, 2, 1), w, b, "silu" if silu else None).transpose(0, 2, 1)
    tol = 2e-5 if kind == "f32" else (1e-2 if kind == "bf16" else 2e-3)
    assert rel_err(out.float().cpu().numpy(), _round(ref, kind)) < tol, (kind, Bsz, Dm, W, L, silu, tables)
