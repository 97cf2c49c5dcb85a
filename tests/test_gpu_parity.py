"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle and the golden vectors.

Tolerances:
  fp32 I/O : the reference's own fp32 bounds (test_selective_scan.py:45 rtol 6e-4 / atol 2e-3;
             test_causal_conv1d.py:31 rtol 3e-4 / atol 1e-3) AND a norm-wise relative error <= 2e-5.
  bf16 I/O : both sides get IDENTICAL bf16 inputs, the oracle computes in fp32 and rounds its result to
             bf16; bound = the reference's bf16 allclose (rtol 3e-2 / atol 5e-2) AND norm-wise <= 1e-3
             (the north-star bar).
"""
import ast

import numpy as np
import pytest
import torch

from conftest import load_golden, rel_err
from oracle import zigma_oracle as zo

pytestmark = pytest.mark.gpu
DEV = "cuda"


def T(a, dtype=torch.float32):
    return None if a is None else torch.from_numpy(np.ascontiguousarray(a)).to(DEV).to(dtype)


def N(t):
    return t.detach().float().cpu().numpy()


@pytest.fixture(scope="module", autouse=True)
def _lib_loaded():
    from zigma_amd import _lib
    _lib.lib()          # fail loudly if the HIP library is missing
    assert torch.cuda.is_available()


# ---------------------------------------------------------------------------------------------------
# selective scan
# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", ["scan_L128", "scan_L1024", "scan_g2_L256", "scan_constBC_L128",
                                  "scan_constB_L128", "scan_plain_L100", "scan_n16_L333"])
def test_scan_generic_vs_golden(name):
    from zigma_amd import _lib
    from zigma_amd.selective_scan_interface import selective_scan_fn
    g = load_golden(name + ".npz")
    out, last = selective_scan_fn(T(g["u"]), T(g["delta"]), T(g["A"]), T(g["B"]), T(g["C"]), T(g.get("D")),
                                  z=T(g.get("z")), delta_bias=T(g.get("delta_bias")),
                                  delta_softplus=bool(g["softplus"]), return_last_state=True)
    assert _lib.last_kernel() == "scan_generic"
    assert np.allclose(N(out), g["out"], rtol=6e-4, atol=2e-3)
    assert np.allclose(N(last), g["last_state"], rtol=6e-4, atol=2e-3)
    assert rel_err(N(out), g["out"]) < 2e-5 and rel_err(N(last), g["last_state"]) < 2e-5


def test_scan_L4096_known_answer_and_chunks():
    """seed-0 recipe of the reference test at L=4096 (2 chunks of 2048): known answers of SURVEY §8c."""
    from zigma_amd.selective_scan_interface import selective_scan_cuda_fwd
    torch.random.manual_seed(0)
    A = -0.5 * torch.rand(4, 8)
    B, C = torch.randn(2, 8, 4096), torch.randn(2, 8, 4096)
    D, z, db = torch.randn(4), torch.randn(2, 4, 4096), 0.5 * torch.rand(4)
    u, delta = torch.randn(2, 4, 4096), 0.5 * torch.rand(2, 4, 4096)
    d = lambda t: t.to(DEV)
    out, x, out_z = selective_scan_cuda_fwd(d(u), d(delta), d(A), d(B).unsqueeze(1), d(C).unsqueeze(1), d(D), d(z), d(db), True)
    assert x.shape == (2, 4, 2, 16) and out.shape == u.shape
    g = load_golden("scan_L4096_sum.npz")
    assert abs(out_z.double().sum().item() - float(g["out_sum"])) < 5e-2
    assert abs(out_z.abs().mean().item() - float(g["out_absmean"])) < 1e-4
    assert np.allclose(N(out_z[0, 0, :3]), g["out_head"], atol=1e-5)
    assert abs(x[:, :, -1, 1::2].double().sum().item() - float(g["state_sum"])) < 1e-3
    # the first chunk's carry equals the state of a scan stopped at 2048
    o2, x2 = selective_scan_cuda_fwd(d(u[..., :2048]).contiguous(), d(delta[..., :2048]).contiguous(), d(A),
                                     d(B[..., :2048]).contiguous().unsqueeze(1), d(C[..., :2048]).contiguous().unsqueeze(1),
                                     d(D), None, d(db), True)
    assert torch.allclose(x[:, :, 0], x2[:, :, 0], rtol=1e-5, atol=1e-6)


def _tok_case(Bsz, L, Di, Nst, dtype, has_z, use_perm, seed, real_A=False):
    rng = np.random.default_rng(seed)
    rnd = zo.bf16_round if dtype == torch.bfloat16 else (lambda a: a.astype(np.float32))
    R = 8
    u = rnd(rng.standard_normal((Bsz, L, Di)).astype(np.float32))
    delta = rnd((rng.random((Bsz, L, Di)) - 0.3).astype(np.float32))
    xdbl = rnd(rng.standard_normal((Bsz, L, R + 2 * Nst)).astype(np.float32))
    zfull = rnd(rng.standard_normal((Bsz, L, 2 * Di)).astype(np.float32))
    A = (-np.exp(np.log(np.arange(1, Nst + 1, dtype=np.float32))[None].repeat(Di, 0)) if real_A
         else -0.5 * rng.random((Di, Nst))).astype(np.float32)
    D = rng.standard_normal(Di).astype(np.float32)
    db = (0.5 * rng.random(Di)).astype(np.float32)
    perm = rng.permutation(L).astype(np.int64) if use_perm else None
    return dict(u=u, delta=delta, xdbl=xdbl, zfull=zfull, A=A, D=D, db=db, perm=perm, R=R, N=Nst, has_z=has_z)


def _run_tok(c, dtype, want_x=False, **extra):
    from zigma_amd.selective_scan_interface import scan_raw
    R, Nst = c["R"], c["N"]
    u, delta, xdbl, zfull = T(c["u"], dtype), T(c["delta"], dtype), T(c["xdbl"], dtype), T(c["zfull"], dtype)
    Bsz, L, Di = u.shape
    perm = None if c["perm"] is None else torch.from_numpy(c["perm"].astype(np.int32)).to(DEV)
    z = zfull[:, :, Di:].transpose(1, 2) if c["has_z"] else None
    y = torch.empty(Bsz, L, Di, device=DEV, dtype=dtype)
    x = torch.empty(Bsz, Di, (L + 2047) // 2048, 2 * Nst, device=DEV) if want_x else None
    kw = dict(out_z=y.transpose(1, 2)) if c["has_z"] else dict(out=y.transpose(1, 2))
    scan_raw(u.transpose(1, 2), delta.transpose(1, 2), T(c["A"]), xdbl[:, :, R:R + Nst].transpose(1, 2).unsqueeze(1),
             xdbl[:, :, R + Nst:].transpose(1, 2).unsqueeze(1), T(c["D"]), z, T(c["db"]), True, x=x,
             z_row_index=perm if c["has_z"] else None, out_row_index=perm, want_out=not c["has_z"], **kw, **extra)
    return y, x


def _oracle_tok(c, dtype):
    R, Nst = c["R"], c["N"]
    Di = c["u"].shape[2]
    tr = lambda a: a.transpose(0, 2, 1)
    z = c["zfull"][:, :, Di:]
    if c["perm"] is not None:
        z = z[:, c["perm"]]
    out, last = zo.selective_scan(tr(c["u"]), tr(c["delta"]), c["A"], tr(c["xdbl"][:, :, R:R + Nst]),
                                  tr(c["xdbl"][:, :, R + Nst:]), c["D"], tr(z) if c["has_z"] else None, c["db"], True,
                                  return_last_state=True)
    out = tr(out)
    if c["perm"] is not None:               # out_tok[perm[k]] = out_scan[k]
        o2 = np.empty_like(out)
        o2[:, c["perm"]] = out
        out = o2
    return (zo.bf16_round(out) if dtype == torch.bfloat16 else out), last


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("L", [1, 2, 15, 16, 17, 100, 1024])
@pytest.mark.parametrize("has_z,use_perm", [(True, True), (True, False), (False, True)])
def test_scan_tok_vs_oracle(dtype, L, has_z, use_perm):
    from zigma_amd import _lib
    c = _tok_case(2, L, 128, 16, dtype, has_z, use_perm, seed=L * 7 + has_z)
    y, x = _run_tok(c, dtype, want_x=True)
    assert _lib.last_kernel().startswith("scan_tok")
    ref, last = _oracle_tok(c, dtype)
    if dtype == torch.float32:
        assert np.allclose(N(y), ref, rtol=6e-4, atol=2e-3)
        assert rel_err(N(y), ref) < 2e-5
    else:
        assert np.allclose(N(y), ref, rtol=3e-2, atol=5e-2)
        assert rel_err(N(y), ref) < 1e-3
    assert rel_err(N(x[:, :, -1, 1::2]), last) < 2e-5


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("use_perm", [True, False])
def test_scan_tok2_hot_kernel_matches_first_generation_and_preactivated_gate(dtype, use_perm, monkeypatch):
    """scan_tok2_kernel (LDS-broadcast B/C, packed math) serves the 16-bit / dstate 16 / gate-only calls: agrees with
    scan_tok_kernel to the last-but-one bit, and with ZIGMA_SCAN_Z_PREACTIVATED it multiplies by z as
    given (silu applied upstream) — checked against the oracle's ungated y times z."""
    from zigma_amd import _lib
    c = _tok_case(3, 64, 192, 16, dtype, True, use_perm, seed=11)
    info = []
    y2, _ = _run_tok(c, dtype, info=info)
    assert info == [_lib.SCAN_KERNEL_TOK2, 0] and _lib.last_kernel() == "scan_tok2_n16"
    info1 = []
    y1, _ = _run_tok(c, dtype, info=info1, _probe_flags=_lib.SCAN_PROBE_V1)
    # same products in the same order; only the sum of the four waves' partials associates differently (fp32), which
    # moves a few 16-bit results by one ulp
    assert info1[0] == _lib.SCAN_KERNEL_TOK and rel_err(N(y2), N(y1)) < 5e-4
    ya, _ = _run_tok(c, dtype, z_preactivated=True)
    Di = c["u"].shape[2]
    zt = N(T(c["zfull"][:, :, Di:], dtype))                     # z as the kernel sees it, token order
    yu, _ = _run_tok(dict(c, has_z=False), dtype)                # ungated y of the first-generation kernel, token order
    assert rel_err(N(ya), N(yu) * zt) < 3e-3                    # (yu is rounded to 16 bits before this product)
    if dtype == torch.bfloat16:
        ref_y, _ = _oracle_tok(dict(c, has_z=False), torch.float32)
        assert rel_err(N(ya), zo.bf16_round(ref_y * zt)) < 1e-3


def _round_to(a, dtype):
    if dtype == torch.bfloat16:
        return zo.bf16_round(a)
    if dtype == torch.float16:
        return a.astype(np.float16).astype(np.float32)
    return a.astype(np.float32)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("L", [16, 32, 48, 1024])
@pytest.mark.parametrize("Di", [64, 192, 1280])
@pytest.mark.parametrize("use_perm", [True, False])
@pytest.mark.parametrize("split", [False, True])
def test_scan_tok2_vs_oracle_sweep(dtype, L, Di, use_perm, split):
    """The HOT kernel (scan_tok2_kernel) against the numpy oracle directly — no first-generation kernel in between — on
    identical 16-bit operands: bf16 and fp16, whole-tile lengths, one / three / twenty slabs, with and without row tables,
    single pass (MODE 0) and the sequence-split form (MODE 1 -> combine -> MODE 2; the chunk carries against the oracle's
    state).  Bounds of the reference's own test (dis_mamba/tests/ops/test_selective_scan.py:45-47: bf16 rtol 3e-2 / atol
    5e-2, fp16 rtol 3e-3 / atol 5e-3) and norm-wise."""
    from zigma_amd import _lib
    if split and L < 32:
        pytest.skip("a split needs two chunks of whole tiles")
    Bsz = 2
    c = _tok_case(Bsz, L, Di, 16, torch.float32, True, use_perm, seed=L + Di + 3 * use_perm, real_A=(Di == 1280))
    for k in ("u", "delta", "xdbl", "zfull"):
        c[k] = _round_to(c[k], dtype)
    info, extra, x = [], {}, None
    if split:
        chunk = 256 if L == 1024 else 16
        x = torch.empty(Bsz, Di, L // chunk, 32, device=DEV, dtype=torch.float32)
        extra = dict(chunk_len=chunk)
    # (_run_tok's own want_x allocates 2048-step carries; the split case passes its own tensor)
    from zigma_amd.selective_scan_interface import scan_raw
    u, delta, xdbl, zfull = (T(c[k], dtype) for k in ("u", "delta", "xdbl", "zfull"))
    R, Nst = c["R"], 16
    perm = None if c["perm"] is None else torch.from_numpy(c["perm"].astype(np.int32)).to(DEV)
    y = torch.empty(Bsz, L, Di, device=DEV, dtype=dtype)
    scan_raw(u.transpose(1, 2), delta.transpose(1, 2), T(c["A"]), xdbl[:, :, R:R + Nst].transpose(1, 2).unsqueeze(1),
             xdbl[:, :, R + Nst:].transpose(1, 2).unsqueeze(1), T(c["D"]), zfull[:, :, Di:].transpose(1, 2), T(c["db"]), True,
             out_z=y.transpose(1, 2), x=x, z_row_index=perm, out_row_index=perm, want_out=False, info=info, **extra)
    assert info[0] == _lib.SCAN_KERNEL_TOK2 and _lib.last_kernel() == "scan_tok2_n16", (info, _lib.last_kernel())
    ref, last = _oracle_tok(c, torch.float32)
    ref = _round_to(ref, dtype)
    rtol, atol, nw = (3e-2, 5e-2, 1e-3) if dtype == torch.bfloat16 else (3e-3, 5e-3, 3e-4)
    assert np.allclose(N(y), ref, rtol=rtol, atol=atol)
    assert rel_err(N(y), ref) < nw, rel_err(N(y), ref)
    if split:
        assert rel_err(N(x[:, :, -1, 1::2]), last) < 2e-5


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("Tq,nseq", [(16, 4), (32, 3), (64, 2)])
def test_scan_tok2_reset_period_vs_oracle(dtype, Tq, nseq):
    """reset_period on the hot kernel (the video temporal layers, mamba_simple.py:420-440 without the transposing copies): a batch
    row is `nseq` independent sequences of Tq steps back to back — against the numpy oracle run on every sequence SEPARATELY (the
    state must restart, nothing may leak across a boundary), with row tables that stay inside each sequence."""
    from zigma_amd import _lib
    from zigma_amd.selective_scan_interface import scan_raw
    Bsz, Di, Nst, R = 3, 128, 16, 8
    L = Tq * nseq
    c = _tok_case(Bsz, L, Di, Nst, torch.float32, True, False, seed=Tq + nseq)
    for k in ("u", "delta", "xdbl", "zfull"):
        c[k] = _round_to(c[k], dtype)
    rng = np.random.default_rng(Tq)
    perm = np.concatenate([q * Tq + rng.permutation(Tq) for q in range(nseq)]).astype(np.int64)
    u, delta, xdbl, zfull = (T(c[k], dtype) for k in ("u", "delta", "xdbl", "zfull"))
    pt = torch.from_numpy(perm.astype(np.int32)).to(DEV)
    y = torch.empty(Bsz, L, Di, device=DEV, dtype=dtype)
    info = []
    scan_raw(u.transpose(1, 2), delta.transpose(1, 2), T(c["A"]), xdbl[:, :, R:R + Nst].transpose(1, 2).unsqueeze(1),
             xdbl[:, :, R + Nst:].transpose(1, 2).unsqueeze(1), T(c["D"]), zfull[:, :, Di:].transpose(1, 2), T(c["db"]), True,
             out_z=y.transpose(1, 2), z_row_index=pt, out_row_index=pt, want_out=False, reset_period=Tq, info=info)
    assert info[0] == _lib.SCAN_KERNEL_TOK2 and _lib.last_kernel() == "scan_tok2_n16"
    ref = np.zeros((Bsz, L, Di), np.float32)
    tr = lambda a: a.transpose(0, 2, 1)
    for q in range(nseq):
        sl = slice(q * Tq, (q + 1) * Tq)
        z = c["zfull"][:, :, Di:][:, perm[sl]]
        o = zo.selective_scan(tr(c["u"][:, sl]), tr(c["delta"][:, sl]), c["A"], tr(c["xdbl"][:, sl, R:R + Nst]), tr(c["xdbl"][:, sl, R + Nst:]),
                              c["D"], tr(z), c["db"], True)
        ref[:, perm[sl]] = tr(o)
    ref = _round_to(ref, dtype)
    assert rel_err(N(y), ref) < (1e-3 if dtype == torch.bfloat16 else 3e-4)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("L,Di,use_perm", [(64, 192, True), (1024, 128, False), (48, 64, True)])
def test_scan_tok2_training_form_vs_oracle(dtype, L, Di, use_perm):
    """The training forward on the hot kernel (MambaInnerFn.forward saves the ungated scan output, selective_scan_interface.py:346-
    365): out_z AND the ungated `out` against the oracle, and the checkpoints (state before every 16-step tile, consumed by
    zigma_selective_scan_bwd) against a float64 recurrence and against the first-generation kernel's."""
    from zigma_amd import _lib
    c = _tok_case(2, L, Di, 16, torch.float32, True, use_perm, seed=L + Di)
    for k in ("u", "delta", "xdbl", "zfull"):
        c[k] = _round_to(c[k], dtype)
    outs = {}
    for tag, fl in (("tok2", 0), ("v1", _lib.SCAN_PROBE_V1)):
        ck = torch.full((2, Di // 64, L // 16, 16, 64), float("nan"), device=DEV)
        out = torch.empty(2, L, Di, device=DEV, dtype=dtype)
        info = []
        y, _ = _run_tok(c, dtype, info=info, out=out.transpose(1, 2), checkpoints=ck, _probe_flags=fl)
        assert info == [_lib.SCAN_KERNEL_TOK2 if tag == "tok2" else _lib.SCAN_KERNEL_TOK, 1], (tag, info)
        outs[tag] = (y, out, ck)
    y, out, ck = outs["tok2"]
    ref_z, _ = _oracle_tok(c, torch.float32)
    ref_o, _ = _oracle_tok(dict(c, has_z=False), torch.float32)
    tol = 1e-3 if dtype == torch.bfloat16 else 3e-4
    assert rel_err(N(y), _round_to(ref_z, dtype)) < tol and rel_err(N(out), _round_to(ref_o, dtype)) < tol
    assert rel_err(N(ck), N(outs["v1"][2])) < 1e-5 and not torch.isnan(ck).any()
    # float64 recurrence for the states before every tile, sample 1, slab 0
    R = c["R"]
    dl = np.log1p(np.exp(np.minimum(c["delta"][1, :, :64].astype(np.float64) + c["db"][:64], 20)))
    dl = np.where(c["delta"][1, :, :64] + c["db"][:64] > 20, c["delta"][1, :, :64] + c["db"][:64], dl)
    h = np.zeros((64, 16))
    for k in range(L):
        if k % 16 == 0:
            assert rel_err(N(ck[1, 0, k // 16]).T, h) < 2e-5 or np.abs(h).max() == 0
        h = np.exp(dl[k][:, None] * c["A"][:64].astype(np.float64)) * h + (dl[k] * c["u"][1, k, :64])[:, None] * c["xdbl"][1, k, R:R + 16][None, :]


@pytest.mark.parametrize("slabs_b,Di", [(2, 64), (33, 64 * 64), (65, 64 * 64)])
def test_scan_tok_state_split_variants(slabs_b, Di):
    """the dispatcher picks 4 / 8 / 16 states per wave by problem size; all must agree with the oracle"""
    from zigma_amd import _lib
    c = _tok_case(slabs_b, 40, Di, 16, torch.bfloat16, True, True, seed=5, real_A=True)
    y, _ = _run_tok(c, torch.bfloat16)
    name = _lib.last_kernel()
    ref, _ = _oracle_tok(c, torch.bfloat16)
    assert rel_err(N(y), ref) < 1e-3, name
    assert np.allclose(N(y), ref, rtol=3e-2, atol=5e-2), name


def test_scan_full_size_properties():
    """BASELINE config 2 size (B=64, Di=1280, L=1024, N=16): generic and token-major kernels agree,
    a sampled slab agrees with the oracle, and the map is linear in u (fp32 I/O)."""
    from zigma_amd.selective_scan_interface import scan_raw
    torch.manual_seed(0)
    Bsz, L, Di, Nst, R = 64, 1024, 1280, 16, 40
    u = torch.randn(Bsz, L, Di, device=DEV)
    u2 = torch.randn(Bsz, L, Di, device=DEV)
    delta = 0.5 * torch.rand(Bsz, L, Di, device=DEV)
    xdbl = torch.randn(Bsz, L, R + 2 * Nst, device=DEV)
    z = torch.randn(Bsz, L, Di, device=DEV)
    A = -torch.exp(torch.log(torch.arange(1, Nst + 1, device=DEV).float())).repeat(Di, 1).contiguous()
    D, db = torch.randn(Di, device=DEV), 0.5 * torch.rand(Di, device=DEV)
    perm = torch.randperm(L, device=DEV).to(torch.int32)
    Bv = xdbl[:, :, R:R + Nst].transpose(1, 2).unsqueeze(1)
    Cv = xdbl[:, :, R + Nst:].transpose(1, 2).unsqueeze(1)

    def run(uu, zz=z):
        y = torch.empty(Bsz, L, Di, device=DEV)
        scan_raw(uu.transpose(1, 2), delta.transpose(1, 2), A, Bv, Cv, D, zz.transpose(1, 2), db, True,
                 out_z=y.transpose(1, 2), z_row_index=perm, out_row_index=perm, want_out=False)
        return y
    y1, y2, y12 = run(u), run(u2), run(u + u2)
    lin = (y12 - (y1 + y2)).norm() / y12.norm()
    assert lin < 1e-5, lin
    # generic kernel on the reference layout (B, D, L) contiguous, permutation applied by torch
    p64 = perm.long()
    uc, dc = u.transpose(1, 2).contiguous(), delta.transpose(1, 2).contiguous()
    zc = z[:, p64].transpose(1, 2).contiguous()
    Bc, Cc = Bv.contiguous(), Cv.contiguous()
    _, oz = scan_raw(uc, dc, A, Bc, Cc, D, zc, db, True, want_out=False)
    ygen = torch.empty_like(y1)
    ygen[:, p64] = oz.transpose(1, 2)
    assert ((ygen - y1).norm() / y1.norm()) < 1e-5
    # oracle on a sampled slab: batches {0, 63}, channels 640..703
    sl = slice(640, 704)
    for b in (0, 63):
        ref = zo.selective_scan(N(u[b:b + 1, :, sl]).transpose(0, 2, 1), N(delta[b:b + 1, :, sl]).transpose(0, 2, 1),
                                N(A[sl]), N(xdbl[b:b + 1, :, R:R + Nst]).transpose(0, 2, 1),
                                N(xdbl[b:b + 1, :, R + Nst:]).transpose(0, 2, 1), N(D[sl]),
                                N(z[b:b + 1][:, p64.cpu()][:, :, sl]).transpose(0, 2, 1), N(db[sl]), True)
        got = N(y1[b:b + 1][:, p64][:, :, sl]).transpose(0, 2, 1)
        assert rel_err(got, ref) < 2e-5


# ---------------------------------------------------------------------------------------------------
# conv, norm
# ---------------------------------------------------------------------------------------------------
def test_conv_vs_golden():
    from zigma_amd.causal_conv1d_interface import causal_conv1d_fn
    g = load_golden("conv.npz")
    for i in range(int(g["n"])):
        act = "silu" if int(g[f"act{i}"]) else None
        out = causal_conv1d_fn(T(g[f"x{i}"]), T(g[f"w{i}"]), T(g.get(f"b{i}")), act)
        assert np.allclose(N(out), g[f"out{i}"], rtol=3e-4, atol=1e-3)
        assert rel_err(N(out), g[f"out{i}"]) < 1e-5
        # channel-last view of the same data -> token-major kernel when dim % 4 == 0
        xl = T(g[f"x{i}"]).transpose(1, 2).contiguous().transpose(1, 2)
        out2 = causal_conv1d_fn(xl, T(g[f"w{i}"]), T(g.get(f"b{i}")), act)
        assert rel_err(N(out2), g[f"out{i}"]) < 1e-5


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("L,W", [(1, 4), (3, 4), (37, 3), (256, 4), (1024, 2)])
def test_conv_tok_gather_vs_oracle(dtype, L, W):
    from zigma_amd import _lib
    from zigma_amd.causal_conv1d_interface import causal_conv1d_raw
    rng = np.random.default_rng(L + W)
    rnd = zo.bf16_round if dtype == torch.bfloat16 else (lambda a: a.astype(np.float32))
    Bsz, Di = 3, 136
    xz = rnd(rng.standard_normal((Bsz, L, 2 * Di)).astype(np.float32))
    w, b = rnd(rng.standard_normal((Di, W)).astype(np.float32)), rnd(rng.standard_normal(Di).astype(np.float32))
    perm = rng.permutation(L)
    xzt = T(xz, dtype)
    out = torch.empty(Bsz, L, Di, device=DEV, dtype=dtype)
    causal_conv1d_raw(xzt[:, :, :Di].transpose(1, 2), T(w, dtype), T(b, dtype), True, out=out.transpose(1, 2),
                      x_row_index=torch.from_numpy(perm.astype(np.int32)).to(DEV))
    assert _lib.last_kernel() == "conv_tok"
    ref = zo.causal_conv1d(xz[:, perm, :Di].transpose(0, 2, 1), w, b, "silu").transpose(0, 2, 1)
    if dtype == torch.bfloat16:
        assert rel_err(N(out), zo.bf16_round(ref)) < 1e-3
        assert np.allclose(N(out), zo.bf16_round(ref), rtol=1e-2, atol=5e-2)
    else:
        assert rel_err(N(out), ref) < 1e-5


def test_norm_vs_golden_and_fused():
    from zigma_amd.layernorm import block_norm, layer_norm_fn, rms_norm_fn
    g = load_golden("norm.npz")
    y, res = rms_norm_fn(T(g["x"]), T(g["w"]), None, residual=T(g["r"]), prenorm=True, residual_in_fp32=True, eps=1e-5)
    assert rel_err(N(y), g["y_rms"]) < 1e-6 and np.array_equal(N(res), g["res_rms"])
    y, res = layer_norm_fn(T(g["x"]), T(g["w"]), T(g["b"]), residual=T(g["r"]), eps=1e-6, prenorm=True)
    assert rel_err(N(y), g["y_ln"]) < 1e-6 and np.array_equal(N(res), g["res_ln"])
    y = rms_norm_fn(T(g["x"]), T(g["w"]), None, eps=1e-5)
    assert rel_err(N(y), g["y_rms0"]) < 1e-6
    # fused block form vs oracle composition, bf16 activations / fp32 residual, E = 640
    rng = np.random.default_rng(0)
    Bsz, L, E = 3, 50, 640
    bf = zo.bf16_round
    x, br = bf(rng.standard_normal((Bsz, L, E)).astype(np.float32)), bf(rng.standard_normal((Bsz, L, E)).astype(np.float32))
    r = rng.standard_normal((Bsz, L, E)).astype(np.float32)
    mod = bf(rng.standard_normal((Bsz, 6 * E)).astype(np.float32) * 0.5)
    w = bf(1 + 0.1 * rng.standard_normal(E).astype(np.float32))
    modt = T(mod, torch.bfloat16)
    xe, res, n, ym = block_norm(T(x, torch.bfloat16), T(w, torch.bfloat16), None, T(r), 1e-5, True, branch=T(br, torch.bfloat16),
                                gate=modt[:, 2 * E:3 * E], shift=modt[:, 0:E], scale=modt[:, E:2 * E], want_x=True)
    xe_ref = bf(x + mod[:, None, 2 * E:3 * E] * br)
    n_ref, res_ref = zo.fused_add_norm(xe_ref, w, None, r, 1e-5, True, True)
    ym_ref = bf(n_ref) * (1 + mod[:, None, E:2 * E]) + mod[:, None, 0:E]
    assert rel_err(N(xe), xe_ref) < 1e-6
    assert rel_err(N(res), res_ref) < 1e-6
    assert rel_err(N(n), bf(n_ref)) < 1e-3 and rel_err(N(ym), bf(ym_ref)) < 1e-3


@pytest.mark.parametrize("M,Di,R,S", [(1, 64, 40, 72), (100, 128, 40, 72), (4096, 1280, 40, 72), (333, 192, 8, 40), (64, 64, 48, 48)])
def test_dt_proj_softplus_mfma_vs_oracle(M, Di, R, S):
    """out = softplus(x[:, :R] @ W.T + b): fp32 oracle on the same bf16 operands, result rounded to bf16."""
    from zigma_amd import _lib
    from zigma_amd.selective_scan_interface import dt_proj_eligible, dt_proj_softplus
    rng = np.random.default_rng(M + R)
    x = zo.bf16_round(rng.standard_normal((2, M, S)).astype(np.float32))
    w = zo.bf16_round((rng.standard_normal((Di, R)) * 0.3).astype(np.float32))
    b = (rng.standard_normal(Di) * 2).astype(np.float32)
    b[0] = 25.0                                   # softplus pass-through branch (x > 20)
    wt = torch.zeros(Di, 48, device=DEV, dtype=torch.bfloat16)[:, :R]
    wt.copy_(T(w, torch.bfloat16))
    xt = T(x, torch.bfloat16)
    assert dt_proj_eligible(xt, R, wt)
    out = dt_proj_softplus(xt, R, wt, T(b), True)
    assert _lib.last_kernel() == "dt_proj_softplus_mfma" and out.shape == (2, M, Di)
    ref = zo.bf16_round(zo.softplus(x[:, :, :R] @ w.T + b))
    assert np.allclose(N(out), ref, rtol=1e-2, atol=1e-2) and rel_err(N(out), ref) < 1e-3
    lin = dt_proj_softplus(xt, R, wt, None, False)
    assert rel_err(N(lin), zo.bf16_round(x[:, :, :R] @ w.T)) < 1e-3


@pytest.mark.parametrize("Bsz,L,Di,R,use_perm", [(2, 64, 128, 40, False), (3, 1024, 192, 40, True), (2, 256, 64, 48, True),
                                                 (5, 48, 1280, 40, True), (1, 16, 64, 32, False), (64, 32, 1536, 48, True), (22, 48, 4096, 40, False)])
@pytest.mark.parametrize("io", ["bf16", "f16"])
def test_scan_tok2_dt_proj_in_kernel_vs_oracle(Bsz, L, Di, R, use_perm, io):
    """ABI 9: dt_proj + bias + softplus inside scan_tok2_kernel (zigma_scan_params_t.dt_x / dt_w) against the numpy oracle's scan on
    delta' = x_dbl[:, :, :R] @ W_dt^T (reference selective_scan_interface.py:323 + softplus(delta + bias) of its kernel) evaluated in
    fp32 on the same bf16 operands, and against the two-kernel path (dt_proj_softplus_kernel + scan) whose delta is a bf16 tensor."""
    from zigma_amd import _lib
    from zigma_amd.selective_scan_interface import dt_in_scan_eligible, dt_proj_softplus, scan_raw
    Nst = 16
    tdt = torch.bfloat16 if io == "bf16" else torch.float16
    bf = zo.bf16_round if io == "bf16" else (lambda a: np.asarray(a, np.float32).astype(np.float16).astype(np.float32))
    rng = np.random.default_rng(Bsz * 1000 + L + R)
    u = bf(rng.standard_normal((Bsz, L, Di)).astype(np.float32))
    z = bf(rng.standard_normal((Bsz, L, Di)).astype(np.float32))
    xd = bf(rng.standard_normal((Bsz, L, R + 2 * Nst)).astype(np.float32))
    w = bf((rng.standard_normal((Di, R)) * R ** -0.5).astype(np.float32))
    A = -np.exp(np.log(np.arange(1, Nst + 1, dtype=np.float32))[None].repeat(Di, 0) + 0.2 * rng.standard_normal((Di, Nst))).astype(np.float32)
    D = (1 + 0.2 * rng.standard_normal(Di)).astype(np.float32)
    db = (rng.standard_normal(Di) - 3.0).astype(np.float32)
    db[0] = 26.0                                                  # softplus pass-through branch (> 20)
    perm = np.random.default_rng(9).permutation(L).astype(np.int64) if use_perm else None
    ut, zt, xt = T(u, tdt), T(z, tdt), T(xd, tdt)
    wt = T(w, tdt)
    pt = None if perm is None else torch.from_numpy(perm).to(DEV).to(torch.int32)
    assert dt_in_scan_eligible(ut, xt, wt)
    Bv, Cv = xt[:, :, R:R + Nst].transpose(1, 2).unsqueeze(1), xt[:, :, R + Nst:].transpose(1, 2).unsqueeze(1)
    y = torch.empty(Bsz, L, Di, device=DEV, dtype=tdt)
    info = []
    scan_raw(ut.transpose(1, 2), None, T(A), Bv, Cv, T(D), zt.transpose(1, 2), T(db), True, out_z=y.transpose(1, 2),
             z_row_index=pt, out_row_index=pt, want_out=False, dt_x=xt, dt_w=wt, info=info)
    # (1536 / 1408 workgroups: one round of SIX resident workgroups per CU instead of a round of five and a tail — the R6 form, round 5)
    r6 = -(-Bsz * (Di // 64) // 1536) < -(-Bsz * (Di // 64) // 1280)
    assert _lib.last_kernel() == ("scan_tok2_n16_dtproj_r6" if r6 else "scan_tok2_n16_dtproj") and info[0] == _lib.SCAN_KERNEL_TOK2
    if r6:      # the five-resident form pinned by the probe bit: same arithmetic, operands fetched one step earlier -> bit-identical
        y5 = torch.empty_like(y)
        scan_raw(ut.transpose(1, 2), None, T(A), Bv, Cv, T(D), zt.transpose(1, 2), T(db), True, out_z=y5.transpose(1, 2),
                 z_row_index=pt, out_row_index=pt, want_out=False, dt_x=xt, dt_w=wt, _probe_flags=1 << _lib.SCAN_PROBE_R5_SHIFT)
        assert _lib.last_kernel() == "scan_tok2_n16_dtproj" and torch.equal(y5, y)
    # oracle: scan position k reads z from row perm[k] and writes row perm[k]
    delta = np.einsum("blr,dr->bdl", xd[:, :, :R].astype(np.float32), w.astype(np.float32))
    zz = z if perm is None else z[:, perm]
    ref = zo.selective_scan(u.transpose(0, 2, 1), delta, A, xd[:, :, R:R + Nst].transpose(0, 2, 1), xd[:, :, R + Nst:].transpose(0, 2, 1),
                            D, zz.transpose(0, 2, 1), db, True).transpose(0, 2, 1)
    got = N(y)
    if perm is not None:
        got = got[:, perm]
    e = rel_err(got, bf(ref))
    assert np.isfinite(got).all() and e < 1e-3, e
    assert np.allclose(got, ref, rtol=3e-2, atol=5e-2)           # the reference's bf16 bounds (test_selective_scan.py:47)
    # the path it replaces
    if io != "bf16":
        return                                                   # (zigma_dt_proj_softplus_fwd is a bf16 kernel)
    dl = dt_proj_softplus(xt, R, wt, T(db), True)
    y2 = torch.empty_like(y)
    scan_raw(ut.transpose(1, 2), dl.transpose(1, 2), T(A), Bv, Cv, T(D), zt.transpose(1, 2), None, False, out_z=y2.transpose(1, 2),
             z_row_index=pt, out_row_index=pt, want_out=False)
    e2 = rel_err(N(y), N(y2))
    print(f"in-kernel dt_proj B={Bsz} L={L} Di={Di} R={R}: vs oracle {e:.2e}; vs dt_proj kernel + scan (bf16 delta) {e2:.2e}")
    assert e2 < 5e-3, e2


@pytest.mark.parametrize("Bsz,L,Di,R", [(2, 64, 128, 40), (3, 1024, 192, 48), (5, 48, 1280, 40)])
def test_scan_tok2_dt_proj_in_kernel_preactivated_gate(Bsz, L, Di, R):
    """dt_proj inside the scan together with ZIGMA_SCAN_Z_PREACTIVATED (round 5: the gate half of in_proj arrives as silu(z)): the kernel
    multiplies by z as it finds it.  Against the oracle's UNGATED y times the same bf16 gate, and against the kernel's own result with the
    raw z (the two differ by the bf16 rounding of silu(z) only)."""
    from zigma_amd import _lib
    from zigma_amd.selective_scan_interface import scan_raw
    Nst = 16
    bf = zo.bf16_round
    rng = np.random.default_rng(Bsz * 77 + L + R)
    u = bf(rng.standard_normal((Bsz, L, Di)).astype(np.float32))
    z = bf(rng.standard_normal((Bsz, L, Di)).astype(np.float32))
    zs = bf(zo.silu(z))
    xd = bf(rng.standard_normal((Bsz, L, R + 2 * Nst)).astype(np.float32))
    w = bf((rng.standard_normal((Di, R)) * R ** -0.5).astype(np.float32))
    A = -np.exp(np.log(np.arange(1, Nst + 1, dtype=np.float32))[None].repeat(Di, 0) + 0.2 * rng.standard_normal((Di, Nst))).astype(np.float32)
    D = (1 + 0.2 * rng.standard_normal(Di)).astype(np.float32)
    db = (rng.standard_normal(Di) - 3.0).astype(np.float32)
    perm = np.random.default_rng(5).permutation(L).astype(np.int64)
    tdt = torch.bfloat16
    ut, zt, zst, xt, wt = T(u, tdt), T(z, tdt), T(zs, tdt), T(xd, tdt), T(w, tdt)
    pt = torch.from_numpy(perm).to(DEV).to(torch.int32)
    Bv, Cv = xt[:, :, R:R + Nst].transpose(1, 2).unsqueeze(1), xt[:, :, R + Nst:].transpose(1, 2).unsqueeze(1)
    y, y_raw = torch.empty(Bsz, L, Di, device=DEV, dtype=tdt), torch.empty(Bsz, L, Di, device=DEV, dtype=tdt)
    scan_raw(ut.transpose(1, 2), None, T(A), Bv, Cv, T(D), zst.transpose(1, 2), T(db), True, out_z=y.transpose(1, 2),
             z_row_index=pt, out_row_index=pt, want_out=False, dt_x=xt, dt_w=wt, z_preactivated=True)
    assert _lib.last_kernel() == "scan_tok2_n16_dtproj"
    scan_raw(ut.transpose(1, 2), None, T(A), Bv, Cv, T(D), zt.transpose(1, 2), T(db), True, out_z=y_raw.transpose(1, 2),
             z_row_index=pt, out_row_index=pt, want_out=False, dt_x=xt, dt_w=wt)
    delta = np.einsum("blr,dr->bdl", xd[:, :, :R].astype(np.float32), w.astype(np.float32))
    ung = zo.selective_scan(u.transpose(0, 2, 1), delta, A, xd[:, :, R:R + Nst].transpose(0, 2, 1), xd[:, :, R + Nst:].transpose(0, 2, 1),
                            D, None, db, True).transpose(0, 2, 1)                  # scan order, no gate
    ref = ung * zs[:, perm]
    e = rel_err(N(y)[:, perm], bf(ref))
    e_raw = rel_err(N(y), N(y_raw))
    print(f"dt_proj in kernel + pre-activated gate B={Bsz} L={L} Di={Di}: vs oracle {e:.2e}; vs the raw-z kernel {e_raw:.2e}")
    assert np.isfinite(N(y)).all() and e < 1e-3, e
    assert e_raw < 4e-3, e_raw


@pytest.mark.parametrize("io", ["bf16", "f16"])
@pytest.mark.parametrize("Bsz,L,Di,R,use_perm", [(70, 64, 192, 40, True), (8, 256, 64 * 26, 48, False), (24, 64, 4096, 48, True)])      # (> 200 workgroups: no sequence split; 24 x 64 = 1536: one round of six per CU)
def test_scan_tok2_accumulating_form(Bsz, L, Di, R, use_perm, io, monkeypatch):
    """ZIGMA_SCAN_ACCUMULATE (round 6): mamba_inner_tok(add_to=y0) — the second sweep of `v2` adding itself to the first one's result in the scan's
    epilogue (reference mamba_simple.py:335-339: out + out_b.flip) — against y0 + mamba_inner_tok(...) evaluated in fp32 on the separate results, on the
    five- and the six-resident form (24 x 64 = 1536 workgroups) of the kernel; the in-place fallback (knob off) gives the rounded-twice sum."""
    import zigma_amd.selective_scan_interface as ssi
    from zigma_amd import _lib
    dtype = torch.bfloat16 if io == "bf16" else torch.float16
    g = torch.Generator().manual_seed(Bsz * L + Di)
    Nst = 16
    xz = torch.randn(Bsz, L, 2 * Di, generator=g).to(DEV, dtype)
    cw, cb = (0.4 * torch.randn(Di, 1, 4, generator=g)).to(DEV, dtype), (0.1 * torch.randn(Di, generator=g)).to(DEV, dtype)
    xw = (Di ** -0.5 * torch.randn(R + 2 * Nst, Di, generator=g)).to(DEV, dtype)
    dw = (R ** -0.5 * torch.randn(Di, R, generator=g)).to(DEV, dtype)
    A = -torch.exp(torch.log(torch.arange(1, Nst + 1).float())[None].repeat(Di, 1) + 0.2 * torch.randn(Di, Nst, generator=g)).to(DEV)
    D, db = (1 + 0.2 * torch.randn(Di, generator=g)).to(DEV), (torch.randn(Di, generator=g) - 3).to(DEV)
    perm = torch.randperm(L, generator=g).to(DEV, torch.int32) if use_perm else None
    y0 = torch.randn(Bsz, L, Di, generator=g).to(DEV, dtype)
    with torch.no_grad():
        y = ssi.mamba_inner_tok(xz, cw, cb, xw, dw, A, D, db, perm=perm)
        k_plain = _lib.last_kernel()
        acc = y0.clone()
        got = ssi.mamba_inner_tok(xz, cw, cb, xw, dw, A, D, db, perm=perm, add_to=acc)
        k_acc = _lib.last_kernel()
        monkeypatch.setattr(ssi, "ACCUMULATE_IN_SCAN", False)
        acc2 = y0.clone()
        got2 = ssi.mamba_inner_tok(xz, cw, cb, xw, dw, A, D, db, perm=perm, add_to=acc2)
    assert got is acc and got2 is acc2
    assert k_plain.startswith("scan_tok2_n16_dtproj") and k_acc == k_plain + "_acc", (k_plain, k_acc)
    assert ("_r6" in k_acc) == (Bsz * (Di // 64) == 1536)
    ref = y0.float() + y.float()                       # y is the rounded result of the plain call: the accumulating form adds the UNROUNDED one
    ulp = 2.0 ** (-8 if io == "bf16" else -11)
    err = (got.float() - ref).abs()
    bound = ulp * (ref.abs() + y.float().abs()) + 1e-6      # one rounding of the sum + the rounding y carried
    assert bool((err <= bound).all()), float((err / bound).max())
    assert torch.equal(got2, (y0 + y))                 # the fallback IS the reference's two-rounding sum
    assert rel_err(N(got), N(got2)) < (4e-3 if io == "bf16" else 6e-4)


@pytest.mark.parametrize("io", ["bf16", "f16"])
@pytest.mark.parametrize("T,nseq,K", [(16, 2, 40), (32, 3, 12), (16, 4, 256)])
def test_dt_proj_in_kernel_with_reset_period(T, nseq, K, io, monkeypatch):
    """The video temporal layers (batch = k pixels, sequence = (b, t) on strided views, conv window and SSM state restarting every T steps) with dt_proj +
    softplus inside the scan kernel (round 6): against the same call on the dt_proj kernel + scan, and against the sequences run as separate batch rows."""
    import zigma_amd.selective_scan_interface as ssi
    from zigma_amd import _lib
    dtype = torch.bfloat16 if io == "bf16" else torch.float16
    g = torch.Generator().manual_seed(T * nseq + K)
    Di, R, Nst = 128, 40, 16
    mk = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc).to(DEV, dtype)
    full = mk(nseq * T, K, 2 * Di)                                  # tokens (b t, k) — the model's (b, (t k), c) layout
    xz = full.transpose(0, 1)                                       # (k, b t, c): a strided view, as Mamba passes it
    cw, cb, xw, dw = mk(Di, 1, 4, sc=0.5), mk(Di, sc=0.1), mk(R + 2 * Nst, Di, sc=Di ** -0.5), mk(Di, R, sc=R ** -0.5)
    A = -torch.exp(0.5 * torch.randn(Di, Nst, generator=g)).to(DEV)
    D, db = torch.randn(Di, generator=g).to(DEV), (torch.randn(Di, generator=g) - 2).to(DEV)
    base = (torch.arange(nseq, dtype=torch.int32) * T).repeat_interleave(T)
    perm = (base + torch.randperm(T, generator=g).to(torch.int32).repeat(nseq)).to(DEV)
    with torch.no_grad():
        y_in = ssi.mamba_inner_tok(xz, cw, cb, xw, dw, A, D, db, perm=perm, reset_period=T)
        k_in = _lib.last_kernel()
        monkeypatch.setattr(ssi, "DT_PROJ_IN_SCAN", False)
        y_k = ssi.mamba_inner_tok(xz, cw, cb, xw, dw, A, D, db, perm=perm, reset_period=T)
        k_k = _lib.last_kernel()
        monkeypatch.setattr(ssi, "DT_PROJ_IN_SCAN", True)
        sep = full.view(nseq, T, K, 2 * Di).permute(2, 0, 1, 3).reshape(K * nseq, T, 2 * Di).contiguous()      # (k b, t, c): every sequence its own row
        y_sep = ssi.mamba_inner_tok(sep, cw, cb, xw, dw, A, D, db, perm=perm[:T])
    assert k_in == "scan_tok2_n16_dtproj" and k_k == "scan_tok2_n16", (k_in, k_k)
    tol = 6e-3 if io == "bf16" else 1e-3                            # (the kernel path rounds delta to the I/O type, the in-kernel one does not)
    assert rel_err(N(y_in), N(y_k)) < tol, rel_err(N(y_in), N(y_k))
    assert rel_err(N(y_in), N(y_sep.view(K, nseq * T, Di))) < tol
    assert torch.isfinite(y_in).all()


def test_scan_dt_in_kernel_limits():
    """What the in-kernel dt_proj (zigma_scan_params_t.dt_x) refuses — the limits tok2_dtp_ok() states, each hit on its own: with dt_x
    set no other kernel serves the call, so the C side answers ZIGMA_ERR_UNSUPPORTED and the caller has to keep the dt_proj kernel."""
    from zigma_amd.selective_scan_interface import dt_in_scan_eligible, scan_raw
    Bsz, L, Di, R, Nst = 1, 32, 64, 40, 16
    ut = torch.randn(Bsz, L, Di, device=DEV).bfloat16()
    xt = torch.randn(Bsz, L, R + 2 * Nst, device=DEV).bfloat16()
    wt = torch.randn(Di, R, device=DEV).bfloat16()
    A = -torch.rand(Di, Nst, device=DEV)
    Bv, Cv = xt[:, :, R:R + Nst].transpose(1, 2).unsqueeze(1), xt[:, :, R + Nst:].transpose(1, 2).unsqueeze(1)
    ok = lambda **kw: scan_raw(ut.transpose(1, 2), None, A, Bv, Cv, None, ut.transpose(1, 2), None, True,
                               **{"dt_x": xt, "dt_w": wt, "want_out": False, **kw})
    ok()                                    # the base case is served (so every refusal below is the one limit it names)
    assert dt_in_scan_eligible(ut, xt, wt, dstate=Nst, z=ut)
    with pytest.raises(RuntimeError):       # delta=None without the pair
        scan_raw(ut.transpose(1, 2), None, A, Bv, Cv, None, ut.transpose(1, 2), None, True)
    with pytest.raises(RuntimeError):       # dt_rank < 32
        ok(dt_w=wt[:, :16].contiguous())
    with pytest.raises(RuntimeError):       # dt_rank % 8 != 0 (36 columns of the 40-wide rows)
        ok(dt_w=wt[:, :36])
    assert not dt_in_scan_eligible(ut, xt, wt[:, :36], dstate=Nst)
    ok(reset_period=16)                     # (round 6: the video temporal layers are served too — test_dt_proj_in_kernel_with_reset_period)
    assert dt_in_scan_eligible(ut, xt, wt, reset_period=16, dstate=Nst) and not dt_in_scan_eligible(ut, xt, wt, reset_period=8, dstate=Nst)
    with pytest.raises(RuntimeError):       # a reset period that is not whole tiles
        ok(reset_period=8)
    with pytest.raises(RuntimeError):       # a carry buffer (sequence split) together with dt_x
        ok(x=torch.empty(Bsz, Di, 2, 2 * Nst, device=DEV), chunk_len=16)
    with pytest.raises(RuntimeError):       # the training form (ungated out) together with dt_x
        ok(out=torch.empty_like(ut).transpose(1, 2), want_out=True)
    # dt_x rows narrower than the 64 columns the two A fragments read
    xn = torch.randn(Bsz, L, 32 + 2 * Nst - 8, device=DEV).bfloat16()        # 56 columns: dt_rank 32 fits, the second fragment does not
    with pytest.raises(RuntimeError):
        scan_raw(ut.transpose(1, 2), None, A, xn[:, :, 24:40].transpose(1, 2).unsqueeze(1), xn[:, :, 40:56].transpose(1, 2).unsqueeze(1), None,
                 ut.transpose(1, 2), None, True, want_out=False, dt_x=xn, dt_w=wt[:, :32].contiguous())
    assert not dt_in_scan_eligible(ut, xn, wt[:, :32].contiguous(), dstate=Nst)
    # dstate != 16: the hot kernel is a 4 waves x 4 states kernel
    A8 = -torch.rand(Di, 8, device=DEV)
    x8 = torch.randn(Bsz, L, 64, device=DEV).bfloat16()
    with pytest.raises(RuntimeError):
        scan_raw(ut.transpose(1, 2), None, A8, x8[:, :, 40:48].transpose(1, 2).unsqueeze(1), x8[:, :, 48:56].transpose(1, 2).unsqueeze(1), None,
                 ut.transpose(1, 2), None, True, want_out=False, dt_x=x8, dt_w=wt)
    assert not dt_in_scan_eligible(ut, x8, wt, dstate=8)
    # fp16 operands ARE served (the f16 form of the MFMA): B / C as strided views of the fp16 x_dbl rows, like the model passes them
    xh, uh = xt.half(), ut.half()
    scan_raw(uh.transpose(1, 2), None, A, xh[:, :, R:R + Nst].transpose(1, 2).unsqueeze(1), xh[:, :, R + Nst:].transpose(1, 2).unsqueeze(1), None,
             uh.transpose(1, 2), None, True, want_out=False, dt_x=xh, dt_w=wt.half())


@pytest.mark.parametrize("d_state", [8, 16, 32])
def test_mamba_inner_tok_other_state_sizes(d_state):
    """with dt_proj inside the scan on by default, a bf16 Mamba with d_state 8 or 32 (dt_rank 32..64) must not reach the
    hot kernel's dt_x form — the gate mirrors tok2_layout_ok (dstate == 16) — and still has to run (dt_proj kernel + first-generation /
    generic scan) and agree with the oracle."""
    from oracle import zigma_oracle as zo
    from zigma_amd.selective_scan_interface import mamba_inner_tok
    from zigma_amd import _lib
    torch.manual_seed(3)
    Bsz, L, Di, R = 2, 64, 128, 48
    xz = torch.randn(Bsz, L, 2 * Di, device=DEV).bfloat16()
    cw, cb = (0.4 * torch.randn(Di, 1, 4, device=DEV)).bfloat16(), (0.1 * torch.randn(Di, device=DEV)).bfloat16()
    xw = (torch.randn(R + 2 * d_state, Di, device=DEV) / Di ** 0.5).bfloat16()
    dw = (torch.randn(Di, R, device=DEV) / R ** 0.5).bfloat16()
    A = -torch.exp(torch.log(torch.arange(1, d_state + 1, device=DEV).float()).repeat(Di, 1)).contiguous()
    D, db = torch.randn(Di, device=DEV), torch.rand(Di, device=DEV) - 2.0
    perm = torch.randperm(L, device=DEV).to(torch.int32)
    y = mamba_inner_tok(xz, cw, cb, xw, dw, A, D, db, perm=perm)
    name = _lib.last_kernel()
    assert ("dtproj" in name) == (d_state == 16), name
    f = lambda t: t.float().cpu().numpy().astype(np.float64)
    bf = lambda a: zo.bf16_round(np.asarray(a, np.float32)).astype(np.float64)
    pn = perm.cpu().numpy()
    xs = f(xz)[:, pn, :Di].transpose(0, 2, 1)                                  # (B, Di, L) in scan order
    u = bf(zo.causal_conv1d(xs, f(cw).reshape(Di, 4), f(cb), activation="silu", dt=np.float64))
    x_dbl = bf(np.einsum("bdl,nd->bln", u, f(xw)))
    delta = np.einsum("blr,dr->bdl", x_dbl[:, :, :R], f(dw))
    Bm, Cm = x_dbl[:, :, R:R + d_state].transpose(0, 2, 1), x_dbl[:, :, R + d_state:].transpose(0, 2, 1)
    z = f(xz)[:, pn, Di:].transpose(0, 2, 1)
    ref = zo.selective_scan(u, delta, f(A), Bm, Cm, f(D), z, f(db), True, dt=np.float64)   # (B, Di, L), scan order
    got = f(y)[:, pn].transpose(0, 2, 1)
    e = rel_err(got, ref)
    assert np.isfinite(got).all() and e < 1e-2, e          # (bf16 delta on the dt_proj-kernel route: 2^-9 on every step size)


# ---------------------------------------------------------------------------------------------------
# mamba inner + model
# ---------------------------------------------------------------------------------------------------
def test_mamba_inner_fn_vs_golden():
    from zigma_amd.selective_scan_interface import mamba_inner_fn
    g = load_golden("mamba_inner.npz")
    out = mamba_inner_fn(T(g["xz"]), T(g["conv_w"]), T(g["conv_b"]), T(g["x_proj_w"]), T(g["dt_proj_w"]),
                         T(g["out_proj_w"]), T(g["out_proj_b"]), T(g["A"]), None, None, T(g["D"]),
                         delta_bias=T(g["delta_bias"]), delta_softplus=True)
    assert rel_err(N(out), g["out"]) < 2e-5
    assert np.allclose(N(out), g["out"], rtol=6e-4, atol=2e-3)


MODEL_FIXTURES = ["zigma_text_zigzag2", "zigma_uncond_zigzag8", "zigma_class_v2", "zigma_hilbert2", "zigma_video_sst"]


def _load_model(name, dtype=torch.float32):
    from zigma_amd.model_zigma import ZigMa
    g = load_golden(name + ".npz")
    cfg = ast.literal_eval(str(g["cfg"]))
    m = ZigMa(device=DEV, dtype=dtype, **cfg).eval()
    sd = {k[3:]: torch.from_numpy(v) for k, v in g.items() if k.startswith("sd.")}
    missing, unexpected = m.load_state_dict(sd, strict=True)
    y = g.get("y")
    if y is not None:
        y = torch.from_numpy(y).to(DEV)
        y = y.long() if cfg.get("num_classes", -1) > 0 else y.to(dtype)
    return m, g, cfg, y


@pytest.mark.parametrize("name", MODEL_FIXTURES)
def test_model_fp32_vs_reference_golden(name):
    m, g, cfg, y = _load_model(name)
    with torch.no_grad():
        out = m(T(g["x"]), T(g["t"]), y)
    assert out.shape == g["out"].shape
    assert rel_err(N(out), g["out"]) < 1e-4, rel_err(N(out), g["out"])


@pytest.mark.parametrize("name", ["zigma_text_zigzag2", "zigma_uncond_zigzag8"])
def test_model_bf16_vs_reference_golden(name):
    """bf16 parameters + activations against the reference's fp32 result: bounded by bf16 resolution of the
    18-deep residual stack; measured ~1e-2, bound 4e-2 (the reference's own bf16 run deviates by the same order)."""
    m, g, cfg, y = _load_model(name, torch.bfloat16)
    with torch.no_grad():
        out = m(T(g["x"]), T(g["t"]), y)
    assert out.dtype == torch.float32           # fp32 in -> fp32 out (cast at the model boundary)
    assert rel_err(N(out), g["out"]) < 4e-2, rel_err(N(out), g["out"])


@pytest.mark.parametrize("name", ["zigma_text_zigzag2", "zigma_video_sst"])
def test_model_fp16_vs_reference_golden(name):
    """fp16 parameters + activations (the f16 instantiations of the scan / conv / norm kernels; dt_proj, x_proj and the
    attention core fall back to the library because their MFMA kernels are bf16-only): bounded by fp16 resolution."""
    m, g, cfg, y = _load_model(name, torch.float16)
    with torch.no_grad():
        out = m(T(g["x"]), T(g["t"]), y)
    assert out.dtype == torch.float32 and torch.isfinite(out).all()
    assert rel_err(N(out), g["out"]) < 1e-2, rel_err(N(out), g["out"])


def test_block_forward_public_api_matches_fused():
    m, g, cfg, y = _load_model("zigma_text_zigzag2")
    torch.manual_seed(0)
    x = torch.randn(2, 64, 32, device=DEV)
    c = torch.randn(2, 32, device=DEV)
    text = torch.randn(2, 5, 32, device=DEV)
    with torch.no_grad():
        h, res = m.blocks[0](x, None, c=c, text=text)
        h2, res2 = m.blocks[1](h, res, c=c, text=text)
    assert h2.shape == x.shape and res2.dtype == torch.float32 and torch.isfinite(h2).all()


def test_sampler_euler_vs_oracle():
    from zigma_amd.transport import Sampler, create_transport
    m, g, cfg, y = _load_model("zigma_uncond_zigzag8")
    state = {k[3:]: v for k, v in g.items() if k.startswith("sd.")}
    om = zo.ZigMaOracle(state, cfg)
    torch.manual_seed(1)
    z0 = torch.randn(2, 4, 8, 8, device=DEV)
    fn = Sampler(create_transport()).sample_ode(sampling_method="euler", num_steps=6)
    with torch.no_grad():
        traj = fn(z0, m.forward)
    ref = zo.sample_ode_fixed(lambda x, t: om.forward(x, t), N(z0), num_steps=6, method="euler")
    assert traj.shape == (6, 2, 4, 8, 8)
    assert rel_err(N(traj[-1]), ref[-1]) < 2e-4


def test_sampler_sde_gpu_matches_oracle_model_on_same_noise():
    """Sampler.sample_sde on the HIP model vs the same sampler driven by the numpy oracle model, same CPU noise stream
    (the Wiener increments come from the CPU generator in both runs): Euler-Maruyama + Mean last step, sigma diffusion."""
    from zigma_amd.transport import ModelType, PathType, Sampler, Transport, WeightType
    m, g, cfg, y = _load_model("zigma_uncond_zigzag8")
    state = {k[3:]: v for k, v in g.items() if k.startswith("sd.")}
    om = zo.ZigMaOracle(state, cfg)
    tr = Transport(model_type=ModelType.VELOCITY, path_type=PathType.LINEAR, loss_type=WeightType.NONE, train_eps=1e-3,
                   sample_eps=1e-3)
    fn = Sampler(tr).sample_sde(sampling_method="Euler", diffusion_form="sigma", diffusion_norm=0.5, num_steps=5)
    torch.manual_seed(3)
    z0 = torch.randn(2, 4, 8, 8)
    torch.manual_seed(4)
    with torch.no_grad():
        xs = fn(z0.to(DEV), m.forward)
    torch.manual_seed(4)
    ref = fn(z0, lambda x, t: torch.from_numpy(om.forward(x.numpy(), t.numpy())))
    assert len(xs) == 5 and xs[-1].is_cuda
    assert rel_err(N(xs[-1]), ref[-1].numpy()) < 5e-4


def test_extension_shims_conventions():
    """out inherits delta's strides, x is (B, D, n_chunks, 2N) f32, errors are RuntimeError."""
    from zigma_amd import extension_shims
    ss, cc = extension_shims.install()
    Bsz, Di, L, Nst = 2, 8, 40, 4
    u = torch.randn(Bsz, Di, L, device=DEV)
    delta = torch.rand(Di, Bsz, L, device=DEV).transpose(0, 1)           # [d][b][l] strides like the reference
    A = -torch.rand(Di, Nst, device=DEV)
    Bm, Cm = torch.randn(Bsz, 1, Nst, L, device=DEV), torch.randn(Bsz, 1, Nst, L, device=DEV)
    out, x = ss.fwd(u, delta, A, Bm, Cm, None, None, None, False)
    assert out.stride() == delta.stride() and x.shape == (Bsz, Di, 1, 2 * Nst) and x.dtype == torch.float32
    ref = zo.selective_scan(N(u), N(delta), N(A), N(Bm), N(Cm))
    assert rel_err(N(out), ref) < 2e-5
    with pytest.raises(RuntimeError):
        ss.fwd(u, delta.half(), A, Bm, Cm, None, None, None, False)
    with pytest.raises(RuntimeError):
        cc.causal_conv1d_fwd(u, torch.randn(Di, 5, device=DEV), None, True)
    with pytest.raises(RuntimeError):
        ss.fwd(u, delta, torch.complex(A, A), Bm, Cm, None, None, None, False)
    with pytest.raises(NotImplementedError):
        cc.causal_conv1d_update()


def test_empty_inputs():
    from zigma_amd.causal_conv1d_interface import causal_conv1d_fn
    from zigma_amd.selective_scan_interface import selective_scan_fn
    u = torch.empty(0, 4, 16, device=DEV)
    out = selective_scan_fn(u, u.clone(), -torch.rand(4, 8, device=DEV), torch.empty(0, 8, 16, device=DEV),
                            torch.empty(0, 8, 16, device=DEV))
    assert out.shape == (0, 4, 16)
    assert causal_conv1d_fn(u, torch.randn(4, 4, device=DEV)).shape == (0, 4, 16)


def test_new_entry_points_reject_what_they_cannot_do():
    """Error behaviour of the round-2 entry points: shapes / strides outside their limits raise (status codes of the C ABI surface as
    RuntimeError), nothing is silently computed another way."""
    from zigma_amd.linear import linear
    from zigma_amd.selective_scan_interface import conv_x_proj
    bf = torch.bfloat16
    x = torch.randn(2, 128, 128, device=DEV, dtype=bf)
    cw, cb = torch.randn(64, 4, device=DEV, dtype=bf), torch.randn(64, device=DEV, dtype=bf)
    with pytest.raises(RuntimeError):                            # n not a multiple of 8
        conv_x_proj(x[:, :, :64], cw, cb, torch.randn(73, 64, device=DEV, dtype=bf))
    with pytest.raises(RuntimeError):                            # seqlen not a multiple of 32
        conv_x_proj(torch.randn(2, 136, 128, device=DEV, dtype=bf)[:, :, :64], cw, cb, torch.randn(72, 64, device=DEV, dtype=bf))
    with pytest.raises(RuntimeError):                            # d_inner not a multiple of 64
        conv_x_proj(torch.randn(2, 128, 96, device=DEV, dtype=bf)[:, :, :48], cw[:48], cb[:48], torch.randn(72, 48, device=DEV, dtype=bf))
    xl = torch.randn(2, 128, 64, device=DEV, dtype=bf)
    with pytest.raises(RuntimeError):                            # gated residual: samples of 128 rows (tiles would straddle samples)
        linear(xl, torch.randn(128, 64, device=DEV, dtype=bf), None, residual=torch.randn(2, 128, 128, device=DEV, dtype=bf),
               gate=torch.randn(2, 128, device=DEV, dtype=bf))


def test_graphed_forward_matches_eager_and_sampler_runs_on_it():
    from zigma_amd.graphs import GraphedForward
    from zigma_amd.transport import Sampler, create_transport
    m, g, cfg, y = _load_model("zigma_text_zigzag2")
    x, t = T(g["x"]), T(g["t"])
    with torch.no_grad():
        ref = m(x, t, y)
    gf = GraphedForward(m, x, t, y)
    out = gf(x, t, y)
    assert torch.equal(out, ref)
    x2 = torch.randn_like(x)
    with torch.no_grad():
        assert torch.equal(gf(x2, t, y), m(x2, t, y))
    fn = Sampler(create_transport()).sample_ode(sampling_method="euler", num_steps=4)
    with torch.no_grad():
        a = fn(x, gf, y=y)
        b = fn(x, m.forward, y=y)
    assert torch.equal(a, b)
    with pytest.raises(RuntimeError):
        gf(x[:1], t[:1], y[:1])


@pytest.mark.parametrize("M,K,Nn", [(1, 256, 72), (300, 1280, 72), (4096, 1536, 80), (257, 512, 96), (64, 256, 40), (16384, 1280, 72), (8192, 1536, 80),
                                    (8192, 1280, 72), (1000, 1024, 96), (16352, 1536, 33), (32768, 1536, 80), (512, 2048, 72)])
def test_x_proj_kernel_vs_oracle(M, K, Nn):
    """x_dbl = u @ W_x^T with the read-bound MFMA kernel (from 16 384 tokens on, or k > 1536) / its split-K form for few tokens (round 5) vs
    float64 numpy on the same bf16 operands (output rounded to bf16); run-to-run identity of the split-K sum."""
    from zigma_amd import _lib
    from zigma_amd.selective_scan_interface import x_proj, x_proj_eligible
    rng = np.random.default_rng(M + K)
    u = zo.bf16_round(rng.standard_normal((M, K)).astype(np.float32))
    w = zo.bf16_round((rng.standard_normal((Nn, K)) * K ** -0.5).astype(np.float32))
    ut, wt = T(u, torch.bfloat16), T(w, torch.bfloat16)
    assert x_proj_eligible(ut, wt) == (M >= 16384 or (M >= 256 and K <= 1536))      # (the policy leaves tiny token counts to the library; the kernel takes them)
    out = x_proj(ut, wt)
    assert _lib.last_kernel() == ("x_proj_splitk" if M < 16384 and K <= 1536 else "x_proj_mfma") and out.shape == (M, Nn)
    assert torch.equal(x_proj(ut, wt), out)
    ref = zo.bf16_round((u.astype(np.float64) @ w.astype(np.float64).T).astype(np.float32))
    assert rel_err(N(out), ref) < 3e-3 and np.allclose(N(out), ref, rtol=2e-2, atol=2e-2)


@pytest.mark.parametrize("Bsz,L,Di,Nn,order,flags", [(2, 128, 64, 72, "id", 0), (1, 256, 192, 40, "rand", 0), (1, 256, 192, 40, "rand", 3),
                                                   (16, 1024, 1280, 72, "rand", 0), (64, 256, 128, 96, "none", 1),
                                                   (4, 4096, 640, 72, "rev", 2), (8, 32, 64, 72, "rand", 0)])
def test_conv_x_proj_kernel_vs_oracle(Bsz, L, Di, Nn, order, flags):
    """The one-pass conv + SiLU + x_proj kernel (flags: 0 = shipped form, 1 / 2 / 3 = three stages / 8-wave workgroups) vs float64 numpy on the same bf16 operands: u (bf16-rounded conv output in scan
    order) and x_dbl = u @ W_x^T evaluated on the kernel's OWN u (the reference rounds u to bf16 before the projection too);
    and against the two separate kernels (same u up to single bf16 roundings of an fp32 sum associated differently)."""
    from zigma_amd import _lib
    from zigma_amd.causal_conv1d_interface import causal_conv1d_raw
    from zigma_amd.selective_scan_interface import conv_x_proj, conv_x_proj_eligible
    rng = np.random.default_rng(L + Di)
    xz = zo.bf16_round(rng.standard_normal((Bsz, L, 2 * Di)).astype(np.float32))       # x is the first half of the in_proj rows
    cw = zo.bf16_round((rng.standard_normal((Di, 4)) * 0.5).astype(np.float32))
    cb = zo.bf16_round((rng.standard_normal(Di) * 0.5).astype(np.float32))
    w = zo.bf16_round((rng.standard_normal((Nn, Di)) * Di ** -0.5).astype(np.float32))
    perm = {"id": np.arange(L), "rand": rng.permutation(L), "rev": np.arange(L)[::-1].copy(), "none": None}[order]
    xzt = T(xz, torch.bfloat16)
    x_half = xzt[:, :, :Di]
    pt = None if perm is None else torch.tensor(perm, device="cuda", dtype=torch.int32)
    cwt, cbt, wt = T(cw, torch.bfloat16), T(cb, torch.bfloat16), T(w, torch.bfloat16)
    assert conv_x_proj_eligible(x_half, cwt, cbt, wt, pt) == (Bsz * L >= 16384)
    u, x_dbl = conv_x_proj(x_half, cwt, cbt, wt, pt, _flags=flags)
    assert _lib.last_kernel() == "conv_x_proj_mfma" and u.shape == (Bsz, L, Di) and x_dbl.shape == (Bsz, L, Nn)
    xg = xz[:, :, :Di].astype(np.float64)
    if perm is not None:
        xg = xg[:, perm]
    xp = np.concatenate([np.zeros((Bsz, 3, Di)), xg], axis=1)
    pre = cb.astype(np.float64) + sum(cw[:, t].astype(np.float64) * xp[:, t:t + L] for t in range(4))
    u_ref = zo.bf16_round((pre / (1.0 + np.exp(-pre))).astype(np.float32))
    assert rel_err(N(u), u_ref) < 3e-3 and np.allclose(N(u), u_ref, rtol=2e-2, atol=2e-2)
    xd_ref = zo.bf16_round((N(u).astype(np.float64) @ w.astype(np.float64).T).astype(np.float32))
    assert rel_err(N(x_dbl), xd_ref) < 3e-3 and np.allclose(N(x_dbl), xd_ref, rtol=2e-2, atol=2e-2)
    u_sep = torch.empty_like(u)
    causal_conv1d_raw(x_half.transpose(1, 2), cwt, cbt, True, out=u_sep.transpose(1, 2), x_row_index=pt)
    differ = (u_sep != u)
    assert differ.float().mean().item() < 0.02
    assert torch.allclose(u_sep.float(), u.float(), rtol=1e-2, atol=1e-3)


@pytest.mark.parametrize("Bsz,L,H,NC", [(2, 100, 8, 77), (1, 64, 3, 128), (3, 17, 8, 5), (2, 256, 8, 81)])
def test_cross_attn_kernel_vs_oracle(Bsz, L, H, NC):
    """softmax(scale Q K^T) V per head on the matrix cores vs a float64 numpy evaluation on the same bf16 operands; K / V
    are row-strided slices of one buffer, as the batched K/V projection hands them over."""
    from zigma_amd import _lib
    from zigma_amd.attention import cross_attn, cross_attn_eligible
    rng = np.random.default_rng(L + NC)
    C = H * 64
    q = zo.bf16_round(rng.standard_normal((Bsz, L, C)).astype(np.float32))
    kv = zo.bf16_round(rng.standard_normal((Bsz, NC, 2, C)).astype(np.float32))
    kvt = T(kv, torch.bfloat16)
    k, v = kvt[:, :, 0], kvt[:, :, 1]
    assert cross_attn_eligible(T(q, torch.bfloat16), k, v, H)
    out = cross_attn(T(q, torch.bfloat16), k, v, H)
    assert _lib.last_kernel() == "cross_attn_mfma" and out.shape == (Bsz, L, C)
    qh = q.reshape(Bsz, L, H, 64).transpose(0, 2, 1, 3).astype(np.float64)
    kh = kv[:, :, 0].reshape(Bsz, NC, H, 64).transpose(0, 2, 1, 3).astype(np.float64)
    vh = kv[:, :, 1].reshape(Bsz, NC, H, 64).transpose(0, 2, 1, 3).astype(np.float64)
    sc = qh @ kh.transpose(0, 1, 3, 2) * 64 ** -0.5
    pr = np.exp(sc - sc.max(-1, keepdims=True))
    ref = ((pr / pr.sum(-1, keepdims=True)) @ vh).transpose(0, 2, 1, 3).reshape(Bsz, L, C)
    assert rel_err(N(out), ref) < 6e-3            # P is rounded to bf16 for the second MFMA, the output to bf16
    assert np.allclose(N(out), ref, rtol=3e-2, atol=3e-2)


def test_dual_stream_forward_matches_plain_forward():
    """two half-batches on two HIP streams inside one hipGraph: same result as the plain forward (samples are
    independent; only the library GEMMs may pick another tile for the smaller M)."""
    from zigma_amd.graphs import DualStreamForward
    m, g, cfg, y = _load_model("zigma_text_zigzag2")
    x, t = T(g["x"]), T(g["t"])
    x4, t4, y4 = torch.cat([x, x.flip(0)]), torch.cat([t, t.flip(0)]), torch.cat([y, y.flip(0)])
    with torch.no_grad():
        ref = m(x4, t4, y4)
    df = DualStreamForward(m, x4, t4, y4, stagger_us=50)
    assert rel_err(N(df(x4, t4, y4)), N(ref)) < 1e-5
    x5 = torch.randn_like(x4)
    with torch.no_grad():
        assert rel_err(N(df(x5, t4, y4)), N(m(x5, t4, y4))) < 1e-5


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("L", [4096, 5000, 8192 + 16])
def test_scan_tok_sequence_split_vs_oracle(dtype, L):
    """few samples + long sequence: the kernel splits the sequence over workgroups (pass 1 carries -> combine ->
    pass 2); result and the carry tensor x must equal the single-pass semantics (oracle)."""
    from zigma_amd import _lib
    c = _tok_case(1, L, 64, 16, dtype, True, True, seed=L, real_A=False)
    y, x = _run_tok(c, dtype, want_x=True)
    assert _lib.last_kernel().startswith("scan_tok")
    ref, last = _oracle_tok(c, dtype)
    if dtype == torch.float32:
        assert rel_err(N(y), ref) < 2e-5
    else:
        assert rel_err(N(y), ref) < 1e-3
    assert rel_err(N(x[:, :, -1, 1::2]), last) < 2e-5
    # carries at the 2048 boundaries == state of a scan stopped there
    c2 = dict(c)
    for k in ("u", "delta", "xdbl", "zfull"):
        c2[k] = c[k][:, :2048]
    c2["perm"] = None
    c2["has_z"] = False
    _, last2048 = _oracle_tok(c2, dtype)
    assert rel_err(N(x[:, :, 0, 1::2]), last2048) < 2e-5


def test_reset_period_matches_separate_sequences():
    """conv + scan with reset_period (independent sequences concatenated along seqlen, read through strided views) vs the
    same sequences run as separate batch rows: the no-copy path of the video temporal layers."""
    from zigma_amd.selective_scan_interface import mamba_inner_tok
    g = torch.Generator(device="cpu").manual_seed(2)
    Bsz, T, K, Di, R, Nst = 3, 16, 8, 128, 8, 16
    mk = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc).to(DEV)
    xz = mk(Bsz, T * K, 2 * Di)                                   # tokens (b, t, k)
    cw, cb, xw, dw = mk(Di, 1, 4, sc=0.5), mk(Di, sc=0.1), mk(R + 2 * Nst, Di, sc=Di ** -0.5), mk(Di, R, sc=R ** -0.5)
    A, D, db = -torch.exp(mk(Di, Nst, sc=0.5)), mk(Di), torch.rand(Di, generator=g).to(DEV)
    perm_t = torch.randperm(T, generator=g).to(DEV, torch.int32)
    out_t = torch.randperm(T, generator=g).to(DEV, torch.int32)
    with torch.no_grad():
        # reference arrangement: (b k) t c with one transposing copy in and one out
        xt = xz.view(Bsz, T, K, 2 * Di).transpose(1, 2).reshape(Bsz * K, T, 2 * Di)
        yt = mamba_inner_tok(xt, cw, cb, xw, dw, A, D, db, perm=perm_t, out_rows=out_t)
        ref = yt.view(Bsz, K, T, Di).transpose(1, 2).reshape(Bsz, T * K, Di)
        # no-copy arrangement
        base = (torch.arange(Bsz, device=DEV, dtype=torch.int32) * T).repeat_interleave(T)
        y = torch.empty(Bsz, T * K, Di, device=DEV)
        mamba_inner_tok(xz.view(Bsz * T, K, 2 * Di).transpose(0, 1), cw, cb, xw, dw, A, D, db, perm=base + perm_t.repeat(Bsz),
                        out_rows=base + out_t.repeat(Bsz), reset_period=T, out=y.view(Bsz * T, K, Di).transpose(0, 1))
    assert rel_err(N(y), N(ref)) < 1e-6


@pytest.mark.parametrize("Bsz,L", [(1, 1024), (2, 400), (3, 272)])
def test_small_batch_sequence_split_matches_single_pass(Bsz, L):
    """serving-size batches: mamba_inner_tok splits the sequence over ~768 workgroups (chunk-local states -> combine ->
    seeded second pass); the result must equal the single-pass kernel's."""
    import zigma_amd.selective_scan_interface as ssi
    g = torch.Generator(device="cpu").manual_seed(L)
    Di, R, Nst = 256, 16, 16
    mk = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc).to(DEV)
    xz = mk(Bsz, L, 2 * Di)
    cw, cb, xw, dw = mk(Di, 1, 4, sc=0.5), mk(Di, sc=0.1), mk(R + 2 * Nst, Di, sc=Di ** -0.5), mk(Di, R, sc=R ** -0.5)
    A, D, db = -torch.exp(mk(Di, Nst, sc=0.5)), mk(Di), torch.rand(Di, generator=g).to(DEV)
    perm = torch.randperm(L, generator=g).to(DEV, torch.int32)
    with torch.no_grad():
        y_split = ssi.mamba_inner_tok(xz, cw, cb, xw, dw, A, D, db, perm=perm)
        ssi.SPLIT_SMALL_BATCH = False
        try:
            y_one = ssi.mamba_inner_tok(xz, cw, cb, xw, dw, A, D, db, perm=perm)
        finally:
            ssi.SPLIT_SMALL_BATCH = True
    assert rel_err(N(y_split), N(y_one)) < 2e-6 and torch.isfinite(y_split).all()


@pytest.mark.parametrize("M,K,N,bias,act", [(4096, 640, 2560, False, 1280), (2048, 1280, 640, False, None), (1040, 640, 512, False, None),
                                           (4096, 512, 640, True, None), (16, 64, 128, True, 64), (272, 128, 384, True, None),
                                           (65536, 640, 2560, False, 1280), (4096, 768, 3072, False, None), (65536, 768, 3072, False, 1536),
                                           (2048, 1536, 768, False, None)])
def test_linear_kernel_vs_float64(M, K, N, bias, act, monkeypatch):
    """zigma_linear_fwd (in_proj / out_proj / to_q / to_out on the matrix cores): every output against a float64 evaluation on
    the same bf16 operands (bias added before the single rounding; SiLU on columns >= act), token counts that are not
    multiples of the 256-token tile, and a strided (sliced) output."""
    from zigma_amd import _lib
    import zigma_amd.routing as zr
    from zigma_amd.linear import linear, linear_eligible
    monkeypatch.setattr(zr, "POLICY", "all")          # (the default policy leaves the 256-wide shapes to the library)
    g = torch.Generator(device="cpu").manual_seed(M + N)
    x = torch.randn(M, K, generator=g).to(DEV, torch.bfloat16)
    w = (torch.randn(N, K, generator=g) * K ** -0.5).to(DEV, torch.bfloat16)
    b = (torch.randn(N, generator=g) * 0.5).to(DEV, torch.bfloat16) if bias else None
    assert linear_eligible(x, w, b)
    y = linear(x, w, b, act)
    assert _lib.last_kernel().startswith("linear_tn_") and y.shape == (M, N)
    rows = torch.arange(M, device=DEV) if M <= 4096 else torch.randint(0, M, (2048,), generator=g).to(DEV)
    ref = x[rows].double() @ w.double().T + (b.double() if bias else 0)
    if act is not None:
        ref[:, act:] = torch.nn.functional.silu(ref[:, act:])
    got = y[rows].double()
    assert float((got - ref).norm() / ref.norm()) < 2.5e-3              # bf16 rounding of the output: 2^-9 rms
    assert torch.allclose(got, ref, rtol=1.6e-2, atol=1e-2)
    if M <= 4096:                                                        # into a column slice of a wider buffer (row pitch != n)
        wide = torch.zeros(M, N + 128, device=DEV, dtype=torch.bfloat16)
        linear(x, w, b, act, out=wide[:, 64:64 + N])
        assert torch.equal(wide[:, 64:64 + N], y) and float(wide[:, :64].abs().max()) == 0 and float(wide[:, 64 + N:].abs().max()) == 0


@pytest.mark.parametrize("M,K,N", [(65536, 640, 2560), (65536, 640, 512), (16384, 192, 4096), (65536, 1280, 256), (32768, 512, 1024),
                                   (65536, 768, 3072), (8192, 768, 3072), (16384, 768, 3072), (65536, 1536, 768), (65536, 768, 512)])      # E = 768: every shipped yaml
def test_linear4w_kernel(M, K, N, monkeypatch):
    """The one-wave-per-SIMD projection kernel (csrc/linear4w.hip, generated main loop): served shapes report it, every output of
    sampled rows against float64 on the same bf16 operands, bit-identity with the 8-wave kernel (same MFMA, same accumulation
    order), run-to-run identity (a synchronisation bug shows as a flicker), and a strided output."""
    from zigma_amd import _lib
    import zigma_amd.routing as zr
    from zigma_amd.linear import linear, linear_eligible
    from zigma_amd.routing import serves_4w as routes_to_4w
    g = torch.Generator(device="cpu").manual_seed(M + N + K)
    x = torch.randn(M, K, generator=g).to(DEV, torch.bfloat16)
    w = (torch.randn(N, K, generator=g) * K ** -0.5).to(DEV, torch.bfloat16)
    assert routes_to_4w(M, N, K)
    monkeypatch.setattr(zr, "POLICY", "all")
    assert linear_eligible(x, w, None)
    y = linear(x, w)
    assert _lib.last_kernel() == "linear4w_256x256" and y.shape == (M, N)
    rows = torch.randint(0, M, (1024,), generator=g).to(DEV)
    rows[:4] = torch.tensor([0, 255, 256, M - 1], device=DEV)
    ref = x[rows].double() @ w.double().T
    got = y[rows].double()
    assert float((got - ref).norm() / ref.norm()) < 2.5e-3
    assert torch.allclose(got, ref, rtol=1.6e-2, atol=1e-2)
    y8 = linear(x, w, _probe_flags=0x2000)
    assert _lib.last_kernel().startswith("linear_tn_")
    assert torch.equal(y, y8)
    for _ in range(4):
        assert torch.equal(linear(x, w), y)
    if N <= 1024:
        wide = torch.zeros(M, N + 256, device=DEV, dtype=torch.bfloat16)
        linear(x, w, out=wide[:, 128:128 + N])
        assert _lib.last_kernel() == "linear4w_256x256"
        assert torch.equal(wide[:, 128:128 + N], y) and float(wide[:, :128].abs().max()) == 0 and float(wide[:, 128 + N:].abs().max()) == 0


@pytest.mark.parametrize("M,K,N", [(65536, 640, 2560), (65536, 640, 512), (32768, 640, 2560), (4096, 640, 8192), (5632, 512, 1024), (512 * 43, 512, 256),
                                   (16384, 640, 1280), (8192, 1280, 640), (16384, 1536, 768), (65536, 1280, 640), (512 * 33, 1536, 128), (8192, 1536, 768)])
def test_linear_ws_kernel(M, K, N):
    """The weight-stationary projection kernel (csrc/linear_ws.hip; in_proj of the default path, reference mamba_simple.py:290-294): every
    output of sampled rows against float64 on the same bf16 operands; the WHOLE result bit-identical with the tiled kernel (same MFMA, same
    accumulation order over k) — which covers every tile, range and panel boundary, odd and even tile counts per workgroup (the epilogue
    rides in the next tile's k-loop; the last one has its own path); run-to-run identity; a strided output; the limits."""
    from zigma_amd import _lib
    from zigma_amd.linear import linear, linear_ws_eligible
    g = torch.Generator(device="cpu").manual_seed(M + N + K)
    x = torch.randn(M, K, generator=g).to(DEV, torch.bfloat16)
    w = (torch.randn(N, K, generator=g) * K ** -0.5).to(DEV, torch.bfloat16)
    assert linear_ws_eligible(x, w)
    y = linear(x, w, weight_stationary=True)
    kname = "linear_ws" if K <= 640 else "linear_ws_128"          # (k = 1280 / 1536: 128-feature panels, one 32-feature block per wave; round 5)
    assert _lib.last_kernel() == kname and y.shape == (M, N)
    rows = torch.randint(0, M, (1024,), generator=g).to(DEV)
    rows[:6] = torch.tensor([0, 63, 64, 511, 512, M - 1], device=DEV)
    ref = x[rows].double() @ w.double().T
    got = y[rows].double()
    assert float((got - ref).norm() / ref.norm()) < 2.5e-3
    assert torch.allclose(got, ref, rtol=1.6e-2, atol=1e-2)
    y8 = linear(x, w, _probe_flags=0x2000)
    assert _lib.last_kernel().startswith("linear_tn_")
    assert torch.equal(y, y8)
    for _ in range(4):
        assert torch.equal(linear(x, w, weight_stationary=True), y)
    wide = torch.zeros(M, N + 256, device=DEV, dtype=torch.bfloat16)
    linear(x, w, out=wide[:, 128:128 + N], weight_stationary=True)
    assert _lib.last_kernel() == kname
    assert torch.equal(wide[:, 128:128 + N], y) and float(wide[:, :128].abs().max()) == 0 and float(wide[:, 128 + N:].abs().max()) == 0
    xs = torch.zeros(M, K + 128, device=DEV, dtype=torch.bfloat16)          # rows of the input 128 elements further apart
    xs[:, :K] = x
    assert torch.equal(linear(xs[:, :K], w, weight_stationary=True), y)


@pytest.mark.parametrize("M,K,N,col", [(65536, 640, 2560, 1280), (16384, 640, 1280, 640), (5632, 512, 1024, 256), (8192, 640, 512, 0)])
def test_linear_ws_silu_epilogue(M, K, N, col):
    """linear_ws_kernel<.., SL>: output columns >= silu_from_col leave as silu(.) of the fp32 accumulator (the pre-activated gate half of in_proj),
    the columns below are bit-identical with the plain kernel; float64 reference on the same bf16 operands; run-to-run identity."""
    from zigma_amd import _lib
    from zigma_amd.linear import linear
    g = torch.Generator(device="cpu").manual_seed(M + N + K + 1)
    x = torch.randn(M, K, generator=g).to(DEV, torch.bfloat16)
    w = (torch.randn(N, K, generator=g) * 2.0 * K ** -0.5).to(DEV, torch.bfloat16)
    y0 = linear(x, w, weight_stationary=True)
    y = linear(x, w, weight_stationary=True, silu_from_col=col)
    assert _lib.last_kernel() == "linear_ws_silu"
    assert torch.equal(y[:, :col], y0[:, :col])
    rows = torch.randint(0, M, (1024,), generator=g).to(DEV)
    rows[:6] = torch.tensor([0, 63, 64, 511, 512, M - 1], device=DEV)
    acc = x[rows].double() @ w.double().T
    ref = torch.nn.functional.silu(acc[:, col:])
    got = y[rows][:, col:].double()
    assert float((got - ref).norm() / ref.norm()) < 2.5e-3
    assert torch.allclose(got, ref, rtol=1.6e-2, atol=1e-2)
    # against silu applied to the ROUNDED product (what the scan's own gate computes from the bf16 z): one bf16 rounding apart
    ref2 = torch.nn.functional.silu(y0[:, col:].float())
    assert float((y[:, col:].float() - ref2).norm() / ref2.norm()) < 4e-3
    assert torch.equal(linear(x, w, weight_stationary=True, silu_from_col=col), y)
    with pytest.raises(RuntimeError):           # a wave's 64 features are all-or-nothing: whole 128-column groups only
        linear(x, w, weight_stationary=True, silu_from_col=col + 64)


@pytest.mark.parametrize("M,K,N", [(8192, 1280, 640), (16384, 1536, 768), (128, 128, 640), (384, 192, 768), (8192, 512, 640), (2048, 1280, 1920), (4096 + 128, 640, 1280), (8192, 640, 512), (16384, 640, 2560)])
def test_linear_sm_kernel(M, K, N):
    """The few-token tiled projection kernel (csrc/linear_sm.hip, round 5: out_proj below the 4-wave kernel's token floor): sampled rows against
    float64 on the same bf16 operands; the WHOLE result bit-identical with the 8-wave tiled kernel (same MFMA, same accumulation order over k:
    covers every tile and wave boundary); run-to-run identity; a strided output; the limits."""
    from zigma_amd import _lib
    from zigma_amd.linear import linear, linear_sm_eligible
    g = torch.Generator(device="cpu").manual_seed(M + N + K + 7)
    x = torch.randn(M, K, generator=g).to(DEV, torch.bfloat16)
    w = (torch.randn(N, K, generator=g) * K ** -0.5).to(DEV, torch.bfloat16)
    assert linear_sm_eligible(x, w)
    y = linear(x, w, few_tokens=True)
    assert _lib.last_kernel() == ("linear_sm_128x160" if N % 160 == 0 else "linear_sm_128x192" if N % 192 == 0 else "linear_sm_128x128") and y.shape == (M, N)
    rows = torch.randint(0, M, (min(M, 1024),), generator=g).to(DEV)
    rows[:4] = torch.tensor([0, 31, 127, M - 1], device=DEV)
    ref = x[rows].double() @ w.double().T
    got = y[rows].double()
    assert float((got - ref).norm() / ref.norm()) < 2.5e-3
    assert torch.allclose(got, ref, rtol=1.6e-2, atol=1e-2)
    y8 = linear(x, w, _probe_flags=0x2000)
    assert _lib.last_kernel().startswith("linear_tn_")
    assert torch.equal(y, y8)
    for _ in range(3):
        assert torch.equal(linear(x, w, few_tokens=True), y)
    wide = torch.zeros(M, N + 256, device=DEV, dtype=torch.bfloat16)
    linear(x, w, out=wide[:, 128:128 + N], few_tokens=True)
    assert torch.equal(wide[:, 128:128 + N], y) and float(wide[:, :128].abs().max()) == 0 and float(wide[:, 128 + N:].abs().max()) == 0
    assert not linear_sm_eligible(x[:104], w) and not linear_sm_eligible(x, w[:96])
    with pytest.raises(RuntimeError):
        linear(x[:104], w, few_tokens=True)


@pytest.mark.parametrize("Bsz,L,K,N,bias,res", [(8, 1024, 1280, 640, False, True), (16, 1024, 512, 640, True, True), (8, 1024, 1536, 768, False, True),
                                                 (4, 512, 512, 640, True, False), (2, 256, 256, 1280, True, True)])
def test_linear_sm_bias_and_gated_residual(Bsz, L, K, N, bias, res):
    """The few-token kernel's epilogue (bias in fp32 before the single rounding; out = residual + gate[b] * bf16(x W^T + bias), reference Block
    model_zigma.py:441-449): bit-identical with the 8-wave tiled kernel's (same arithmetic and rounding points) and against float64 on sampled rows."""
    from zigma_amd import _lib
    from zigma_amd.linear import linear, linear_sm_eligible
    g = torch.Generator(device="cpu").manual_seed(Bsz + K + N + 3)
    x = torch.randn(Bsz, L, K, generator=g).to(DEV, torch.bfloat16)
    w = (torch.randn(N, K, generator=g) * K ** -0.5).to(DEV, torch.bfloat16)
    b = (torch.randn(N, generator=g) * 0.2).to(DEV, torch.bfloat16) if bias else None
    r = torch.randn(Bsz, L, N, generator=g).to(DEV, torch.bfloat16) if res else None
    gt = torch.randn(Bsz, N, generator=g).to(DEV, torch.bfloat16) if res else None
    assert linear_sm_eligible(x, w, b)
    y = linear(x, w, b, residual=r, gate=gt, few_tokens=True)
    assert _lib.last_kernel().startswith("linear_sm_128x")
    y8 = linear(x, w, b, residual=r, gate=gt, _probe_flags=0x2000)
    assert _lib.last_kernel().startswith("linear_tn_")
    assert torch.equal(y, y8)
    v = x.double() @ w.double().T + (b.double() if bias else 0)
    ref = (r.double() + gt.double().unsqueeze(1) * v.bfloat16().double()) if res else v
    assert float((y.double() - ref).norm() / ref.norm()) < 3e-3
    assert torch.equal(linear(x, w, b, residual=r, gate=gt, few_tokens=True), y)


@pytest.mark.parametrize("E", [512, 640, 768, 1024])
@pytest.mark.parametrize("Bsz", [4, 8, 16, 64])
def test_every_routing_cell_vs_float64(E, Bsz):
    """zigma_amd/routing.py: EVERY cell of the table the models can reach — the four projection roles x E in {512, 640, 768, 1024} x 4096 / 8192 / 16 384 /
    65 536 tokens — through linear.project (the one dispatch of the block loop), each result against float64 on the same bf16 operands, incl. bias and the
    gated residual where the role carries them; the kernel that served the call must be the one the table names, and no tensor-level refusal may occur
    (a route nobody compares with anything was VERDICT r5's weak 1)."""
    import zigma_amd.routing as zr
    from zigma_amd import _lib
    from zigma_amd.linear import project
    L = 1024
    tokens = Bsz * L
    g = torch.Generator(device="cpu").manual_seed(E + Bsz)
    zr.REFUSED.clear()
    for role, (n, k) in {"in_proj": (4 * E, E), "out_proj": (E, 2 * E), "to_q": (512, E), "to_out": (E, 512)}.items():
        x = torch.randn(Bsz, L, k, generator=g).to(DEV, torch.bfloat16)
        w = (torch.randn(n, k, generator=g) * k ** -0.5).to(DEV, torch.bfloat16)
        b = (torch.randn(n, generator=g) * 0.3).to(DEV, torch.bfloat16) if role == "to_out" else None
        res = torch.randn(Bsz, L, n, generator=g).to(DEV, torch.bfloat16) if role in ("out_proj", "to_out") else None
        gate = torch.randn(Bsz, n, generator=g).to(DEV, torch.bfloat16) if res is not None else None
        r = zr.route(role, tokens, n, k)
        trace = []
        _lib.TRACE = trace
        try:
            with torch.no_grad():
                y = project(role, x, w, b, residual=res, gate=gate)
        finally:
            _lib.TRACE = None
        served = [kern for fn, kern, _ in trace if fn == "zigma_linear_fwd"]
        if r.kernel == "library":
            assert served == [], (role, E, tokens, r, served)
        else:
            want = zr.kernel_name(r, tokens, n, k)
            assert len(served) == (2 if r.kernel == "tiled_halves" else 1) and all(sv.startswith(want) for sv in served), (role, E, tokens, r, served)
            fused = [bool(P.residual) for fn, _, P in trace if fn == "zigma_linear_fwd"]
            assert fused == [bool(res is not None and r.fuse_add)] * len(served), (role, E, tokens, r, fused)
        rows = torch.randint(0, tokens, (512,), generator=g).to(DEV)
        rows[:3] = torch.tensor([0, L - 1, tokens - 1], device=DEV)
        x2, y2 = x.view(tokens, k), y.view(tokens, n)
        ref = x2[rows].double() @ w.double().T + (b.double() if b is not None else 0)
        if res is not None:
            ref = res.view(tokens, n)[rows].double() + gate.double()[rows // L] * ref
        err = float((y2[rows].double() - ref).norm() / ref.norm())
        assert err < 4e-3, (role, E, tokens, r, err)             # (bf16 rounding of the product and, unfused, of the sum)
    assert zr.REFUSED == [], zr.REFUSED


def test_linear_ws_limits():
    from zigma_amd.linear import linear, linear_ws_eligible
    x = torch.randn(4096, 640, device=DEV).bfloat16()
    for w, xx in ((torch.randn(384, 640, device=DEV).bfloat16(), x),               # n % 256
                  (torch.randn(512, 768, device=DEV).bfloat16(), torch.randn(4096, 768, device=DEV).bfloat16()),      # k > 640
                  (torch.randn(512, 640, device=DEV).bfloat16(), x[:4000])):       # tokens % 512
        assert not linear_ws_eligible(xx, w)
        with pytest.raises(RuntimeError):
            linear(xx, w, weight_stationary=True)
    with pytest.raises(RuntimeError):                                              # no epilogue operands in this kernel
        linear(x, torch.randn(512, 640, device=DEV).bfloat16(), torch.zeros(512, device=DEV).bfloat16(), weight_stationary=True)


@pytest.mark.parametrize("Bsz,L,K,N,bias,res", [(64, 1024, 1280, 640, False, False), (64, 1024, 1280, 640, False, True), (64, 1024, 512, 640, True, True),
                                                 (32, 2048, 192, 384, False, True), (128, 256, 256, 1152, True, True)])
def test_linear4w_narrow_tiles_and_gated_residual(Bsz, L, K, N, bias, res, monkeypatch):
    """The 4-wave kernel on n % 256 == 128 (a 128-wide tile behind the 256-wide ones) and with the block's gated branch add in its
    epilogue (+ bias as a rank-1 MFMA): out = residual + gate[b] * (x W^T + bias) against float64 on the same bf16 operands for
    sampled rows of every sample position class, against the 8-wave kernel (which rounds x W^T + b to bf16 before the gate: the two
    differ by that rounding only), run-to-run identity."""
    from zigma_amd import _lib
    import zigma_amd.routing as zr
    from zigma_amd.linear import linear
    monkeypatch.setattr(zr, "POLICY", "all")
    g = torch.Generator(device="cpu").manual_seed(Bsz + K + N)
    M = Bsz * L
    x = torch.randn(Bsz, L, K, generator=g).to(DEV, torch.bfloat16)
    w = (torch.randn(N, K, generator=g) * K ** -0.5).to(DEV, torch.bfloat16)
    b = (torch.randn(N, generator=g) * 0.5).to(DEV, torch.bfloat16) if bias else None
    r = torch.randn(Bsz, L, N, generator=g).to(DEV, torch.bfloat16) if res else None
    gt = torch.randn(Bsz, N, generator=g).to(DEV, torch.bfloat16) if res else None
    y = linear(x, w, b, residual=r, gate=gt)
    assert _lib.last_kernel() == ("linear4w_256x256+128" if N % 256 else "linear4w_256x256"), _lib.last_kernel()
    rows = torch.randint(0, M, (768,), generator=g).to(DEV)
    rows[:6] = torch.tensor([0, 127, 128, L - 1, L % M, M - 1], device=DEV)
    val = x.view(M, K)[rows].double() @ w.double().T + (b.double() if bias else 0)
    ref = val if not res else r.view(M, N)[rows].double() + gt[rows // L].double() * val
    got = y.view(M, N)[rows].double()
    assert float((got - ref).norm() / ref.norm()) < 2.5e-3
    assert torch.allclose(got, ref, rtol=1.6e-2, atol=2e-2)
    y8 = linear(x, w, b, residual=r, gate=gt, _probe_flags=0x2000)
    assert _lib.last_kernel().startswith("linear_tn_")
    assert float((y.float() - y8.float()).norm() / y8.float().norm()) < 4e-3
    if not res:
        assert torch.equal(y, y8)
    for _ in range(3):
        assert torch.equal(linear(x, w, b, residual=r, gate=gt), y)


@pytest.mark.parametrize("Bsz,L,K,Nn,bias", [(2, 256, 512, 640, True), (3, 512, 128, 128, False), (16, 1024, 512, 640, True)])
def test_linear_gated_residual_epilogue(Bsz, L, K, Nn, bias, monkeypatch):
    """out = residual + gate[b] * bf16(x @ W^T + bias) in the projection kernel's epilogue (the block's gated branch add) vs the
    same thing spelled out in float64 with the projection rounded to bf16 first."""
    import zigma_amd.routing as zr
    from zigma_amd import _lib
    from zigma_amd.linear import gated_residual_eligible, linear
    monkeypatch.setattr(zr, "POLICY", "all")
    g = torch.Generator(device="cpu").manual_seed(L + Nn)
    bf = torch.bfloat16
    x = torch.randn(Bsz, L, K, generator=g).to(DEV, bf)
    w = (torch.randn(Nn, K, generator=g) * K ** -0.5).to(DEV, bf)
    b = (torch.randn(Nn, generator=g) * 0.5).to(DEV, bf) if bias else None
    wide = torch.randn(Bsz, L, Nn + 64, generator=g).to(DEV, bf)
    res = wide[:, :, 64:]                                                        # a column slice: row pitch != n
    gate = torch.randn(Bsz, 3 * Nn, generator=g).to(DEV, bf)[:, Nn:2 * Nn]      # a chunk of the adaLN rows
    assert gated_residual_eligible(x, res, gate)
    out = linear(x, w, b, residual=res, gate=gate)
    assert _lib.last_kernel() == "linear_tn_256x128" and out.shape == (Bsz, L, Nn)
    proj = (x.double() @ w.double().T + (b.double() if bias else 0)).to(bf).double()
    ref = res.double() + gate.double().unsqueeze(1) * proj
    assert float((out.double() - ref).norm() / ref.norm()) < 2.5e-3 and torch.allclose(out.double(), ref, rtol=1.6e-2, atol=1.6e-2)
    plain = linear(x, w, b)                                                      # and the fp32 fma of the kernel's own bf16 projection
    exact = torch.addcmul(res.float(), gate.float().unsqueeze(1), plain.float()).to(bf)
    assert (out != exact).float().mean().item() < 1e-3                           # (fma vs mul + add: ties in the last place)


def test_backward_batch_beyond_the_grid_limit_runs_in_slices():
    """batch > 65535 through the BACKWARD wrappers (scan and conv): slices, parameter gradients summed; compared with the same
    rows run as two ordinary batches."""
    from zigma_amd.causal_conv1d_interface import conv_bwd_tok
    from zigma_amd.selective_scan_interface import scan_bwd_tok
    g = torch.Generator(device="cpu").manual_seed(5)
    Bsz, L, Dm, Nst = 65535 + 9, 16, 64, 16
    bf = torch.bfloat16
    mk = lambda *s, sc=1.0, dt=bf: (torch.randn(*s, generator=g) * sc).to(DEV, dt)
    u, z, dout, out = mk(Bsz, L, Dm), mk(Bsz, L, Dm), mk(Bsz, L, Dm), mk(Bsz, L, Dm)
    delta = (torch.rand(Bsz, L, Dm, generator=g) * 0.5).to(DEV, bf)
    Bm, Cm = mk(Bsz, L, Nst), mk(Bsz, L, Nst)
    A = (-torch.exp(torch.randn(Dm, Nst, generator=g) * 0.5)).to(DEV)
    D, db = mk(Dm, dt=torch.float32), (torch.rand(Dm, generator=g) * 0.5).to(DEV)
    full = scan_bwd_tok(u, delta, A, Bm, Cm, D, z, db, dout, out, True)
    h = Bsz // 2
    p1 = scan_bwd_tok(u[:h], delta[:h], A, Bm[:h], Cm[:h], D, z[:h], db, dout[:h], out[:h], True)
    p2 = scan_bwd_tok(u[h:], delta[h:], A, Bm[h:], Cm[h:], D, z[h:], db, dout[h:], out[h:], True)
    for i in (0, 1, 3, 4, 6):                                   # per-sample outputs: identical rows
        assert torch.equal(full[i][:h], p1[i]) and torch.equal(full[i][h:], p2[i])
    for i in (2, 5, 7):                                         # parameter gradients: sums over the batch (fp32, other order)
        assert rel_err(N(full[i]), N(p1[i] + p2[i])) < 1e-5
    w, cb = mk(Dm, 4, sc=0.5), mk(Dm, sc=0.1)
    perm = torch.randperm(L, generator=g).to(DEV, torch.int32)
    fx, fw, fb = conv_bwd_tok(u, w, cb, dout, True, perm)
    x1, w1, b1 = conv_bwd_tok(u[:h], w, cb, dout[:h], True, perm)
    x2, w2, b2 = conv_bwd_tok(u[h:], w, cb, dout[h:], True, perm)
    assert torch.equal(fx[:h], x1) and torch.equal(fx[h:], x2)
    assert rel_err(N(fw), N(w1 + w2)) < 1e-5 and rel_err(N(fb), N(b1 + b2)) < 1e-5


def test_batch_beyond_the_grid_limit_runs_in_slices():
    """batch > 65535 (the differentiable video temporal path reshapes to (batch * K, T, C)): conv and scan forward slice the
    batch instead of failing the launch; compared with the same rows run as a small batch."""
    from zigma_amd.causal_conv1d_interface import causal_conv1d_raw
    from zigma_amd.selective_scan_interface import scan_raw
    Bsz, L, Di, Nst = 65536 + 40, 16, 64, 16
    g = torch.Generator(device="cpu").manual_seed(3)
    mk = lambda *s_: torch.randn(*s_, generator=g).to(DEV)
    x, w, bias = mk(Bsz, L, Di), mk(Di, 4), mk(Di)
    u = torch.empty(Bsz, L, Di, device=DEV)
    causal_conv1d_raw(x.transpose(1, 2), w, bias, True, out=u.transpose(1, 2))
    sel = torch.tensor([0, 1, 65534, 65535, 65536, Bsz - 1], device=DEV)
    u_ref = torch.empty(len(sel), L, Di, device=DEV)
    causal_conv1d_raw(x[sel].transpose(1, 2), w, bias, True, out=u_ref.transpose(1, 2))
    assert torch.equal(u[sel], u_ref)
    delta, xd, z = 0.5 * torch.rand(Bsz, L, Di, generator=g).to(DEV), mk(Bsz, L, 2 * Nst), mk(Bsz, L, Di)
    A, D = -torch.rand(Di, Nst, generator=g).to(DEV), mk(Di)
    run = lambda s_: scan_raw(u[s_].transpose(1, 2), delta[s_].transpose(1, 2), A, xd[s_][:, :, :Nst].transpose(1, 2).unsqueeze(1),
                              xd[s_][:, :, Nst:].transpose(1, 2).unsqueeze(1), D, z[s_].transpose(1, 2), None, False, want_out=False)[1]
    y = run(slice(None))
    assert torch.equal(y[sel], run(sel))
