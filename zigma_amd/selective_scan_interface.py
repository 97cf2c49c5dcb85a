"""Selective-scan op layer on the HIP kernels.

Mirrors the reference's `dis_mamba/mamba_ssm/ops/selective_scan_interface.py` (same names, argument
order and error behaviour) for the FORWARD path:

    selective_scan_cuda_fwd   <-> selective_scan_cuda.fwd            (selective_scan.cpp:226-336)
    selective_scan_fn         <-> selective_scan_fn                  (selective_scan_interface.py:77-83)
    mamba_inner_fn            <-> mamba_inner_fn / MambaInnerFn.forward        (:296-365, :606-614)
    mamba_inner_fn_no_out_proj<-> MambaInnerFnNoOutProj.forward                (:155-224)

plus `mamba_inner_tok`, the token-major fused form the ZigMa block uses on MI355X: conv, scan and
the zigzag gather/scatter run on (B, L, C) activations straight out of / into the projection GEMMs,
so none of the reference's transposes, `index_select`s or `cat`s exist.

Backward: `scan_bwd_tok` binds zigma_selective_scan_bwd; `MambaInnerTokFn` is the autograd
form of `mamba_inner_tok` (MambaInnerFn.backward, :367-434) that the blocks use when autograd is recording.  The
(B, D, L)-layout entry points (`selective_scan_fn`, `mamba_inner_fn`) stay forward-only.
"""

import torch
import torch.nn.functional as F

from . import _lib
from .causal_conv1d_interface import causal_conv1d_raw, conv_bwd_tok
from .wgrad import wgrad


SPLIT_SMALL_BATCH = True     # tools/latency_probe.py flips this to measure the effect of the small-batch sequence split
# module-level knobs for the A/B tools and tests (tools/, tests/ set them directly; no environment switches)
USE_CONV_X_PROJ = True               # conv + SiLU + x_proj in one kernel (u written once, never read back); False: the two kernels
CONV_X_PROJ_MIN_POSITIONS = 16384    # below: too few workgroups (128 positions each)
DT_PROJ_FLAGS = 0                    # 1: four-byte stores (A/B probe of dt_proj.hip)
USE_X_PROJ_KERNEL = True      # own MFMA kernel for the skinny x_proj instead of the library GEMM (same speed stand-alone)
SPLIT_MAX_WGS = 200          # ... and sweeps this: split when batch * d_inner / 64 is at most this many workgroups (tools/split_threshold_probe.py)


def split_chunk_len(batch, d_inner, seqlen, reset_period=0):
    """Chunk length (a multiple of 16) for the sequence-split mode of the token-major scan, or 0 for a single pass.
    The plain grid is batch * d_inner / 64 workgroups; the split pays (it runs the recurrence twice) only when that
    leaves most of the 256 CUs idle: small batches at L >= 256 (threshold from tools/split_threshold_probe.py), or
    long sequences (L >= 4096: chunks of at most the reference's 2048 steps, sized for ~5 workgroups per CU)."""
    wgs = batch * (d_inner // 64)
    if reset_period or wgs < 1:
        return 0
    if seqlen >= 4096:                                            # long sequences: ~1280 workgroups = 5 per CU, all resident
        if wgs >= 768:
            return 0
        per = -(-seqlen // -(-1280 // wgs))
        return min(2048, max(256, (per + 15) // 16 * 16))
    if not (SPLIT_SMALL_BATCH and wgs <= SPLIT_MAX_WGS and seqlen >= 256):
        return 0
    per = -(-seqlen // -(-768 // wgs))                            # steps per chunk that give ~768 workgroups
    chunk = min(2048, max(32, (per + 15) // 16 * 16))
    return chunk if -(-seqlen // chunk) >= 2 else 0


def _as_bgnl(M, name):
    """(B, N, L) -> (B, 1, N, L) like SelectiveScanFn.forward (:30-35)."""
    if M.dim() == 3:
        return M.unsqueeze(1)
    if M.dim() != 4:
        raise RuntimeError(f"{name} must be (D, N), (B, N, L) or (B, G, N, L)")
    return M


def scan_raw(u, delta, A, B, C, D=None, z=None, delta_bias=None, delta_softplus=False, *, out=None, out_z=None,
             x=None, z_row_index=None, out_row_index=None, want_out=True, checkpoints=None, reset_period=0,
             chunk_len=2048, z_preactivated=False, info=None, _probe_flags=0, dt_x=None, dt_w=None, accumulate=False):
    """Launch zigma_selective_scan_fwd.  All tensors are logical (batch, dim, seqlen) VIEWS with arbitrary
    strides (token-major tensors come in as `.transpose(1, 2)`); B/C are (D, N) f32 or (B, G, N, L) views.
    Outputs that are None are allocated here with the reference's conventions (out like delta, out_z like z).
    z_preactivated: z already holds silu(z) (ZIGMA_SCAN_Z_PREACTIVATED; hot token-major kernel only).
    info: optional list; receives [kernel family (_lib.SCAN_KERNEL_*), 1 if `checkpoints` is being written].
    dt_x, dt_w (ABI 9, with delta=None): dt_proj + bias + softplus inside the token-major hot kernel — dt_x (batch, seqlen, >= dt_rank)
    bf16 rows in SCAN order (x_dbl as x_proj wrote it), dt_w (dim, dt_rank); delta' = softplus(dt_x[..., :dt_rank] @ dt_w.T + delta_bias).
    dt_x, dt_w AND delta AND x (ABI 10): the sequence split with dt_proj inside its first pass — delta (uninitialised, shape of u) receives the step sizes.
    accumulate (with dt_x / dt_w only): out_z += y * silu(z) — the second sweep of `v2` adds itself to the first one's result (ZIGMA_SCAN_ACCUMULATE)."""
    dev = _lib.require_device(u, delta, A, B, C, D, z, delta_bias, out, out_z, x, z_row_index, out_row_index, dt_x, dt_w)
    if delta is None:
        if dt_x is None or dt_w is None or not delta_softplus:
            raise RuntimeError("delta=None needs dt_x, dt_w and delta_softplus=True")
    elif dt_x is not None or dt_w is not None:
        # (ABI 10) with the carry tensor x — the sequence split — delta is a WORKSPACE: the first pass forms softplus(dt_proj + bias) itself and
        # writes it there for the second pass; without x the two are alternatives
        if x is None or dt_x is None or dt_w is None or not delta_softplus:
            raise RuntimeError("pass either delta or (dt_x, dt_w); both only with x (sequence split: delta is the workspace the first pass fills)")
    if u.dim() != 3 or (delta is not None and delta.shape != u.shape):
        raise RuntimeError("u and delta must both be (batch, dim, seqlen)")
    if A.is_complex():
        raise RuntimeError("zigma_amd: complex A is out of scope (ZigMa's A is real, mamba_simple.py:298)")
    if A.dtype != torch.float32:
        raise RuntimeError("A must be float32")
    if delta is not None and delta.dtype != u.dtype:
        raise RuntimeError("delta must have the dtype of u")
    batch, dim, L = u.shape
    N = A.shape[1]
    if A.shape[0] != dim:
        raise RuntimeError("A must be (dim, dstate)")
    if N > 256:
        raise RuntimeError("selective_scan only supports state dimension <= 256")
    var_b, var_c = B.dim() >= 3, C.dim() >= 3
    P = _lib.ScanParams()
    P.batch, P.dim, P.seqlen, P.dstate = batch, dim, L, N
    P.delta_softplus = int(bool(delta_softplus))
    P.io_dtype = _lib.dtype_id(u)
    if accumulate and (dt_x is None or out_z is None):
        raise RuntimeError("accumulate=True is served by the in-kernel dt_proj form only (dt_x / dt_w) and needs the out_z to add to")
    P.chunk_len, P.flags = int(chunk_len), (_lib.SCAN_Z_PREACTIVATED if z_preactivated else 0) | (_lib.SCAN_ACCUMULATE if accumulate else 0) | int(_probe_flags)
    P.is_variable_B, P.is_variable_C = int(var_b), int(var_c)
    groups = 1
    bc_dt = None
    for name, M, var in (("B", B, var_b), ("C", C, var_c)):
        if var:
            M = _as_bgnl(M, name)
            if M.shape[0] != batch or M.shape[2] != N or M.shape[3] != L:
                raise RuntimeError(f"{name} must be (batch, groups, dstate, seqlen)")
            groups = M.shape[1]
            if bc_dt is not None and M.dtype != bc_dt:
                raise RuntimeError("variable B and C must share a dtype")
            bc_dt = M.dtype
            sb, sg, sn, sl = M.stride()
            setattr(P, f"{name}_batch_stride", sb), setattr(P, f"{name}_group_stride", sg)
            setattr(P, f"{name}_dstate_stride", sn), setattr(P, f"{name}_l_stride", sl)
        else:
            if M.shape != (dim, N) or M.dtype != torch.float32:
                raise RuntimeError(f"constant {name} must be float32 (dim, dstate)")
            setattr(P, f"{name}_d_stride", M.stride(0)), setattr(P, f"{name}_dstate_stride", M.stride(1))
        setattr(P, name, _lib.ptr(M))
        if name == "B":
            Bk = M
        else:
            Ck = M
    if var_b and var_c and _as_bgnl(B, "B").shape[1] != _as_bgnl(C, "C").shape[1]:
        raise RuntimeError("B and C must have the same number of groups")
    if dim % groups != 0:
        raise RuntimeError("dim must be divisible by the number of B/C groups")
    P.n_groups = groups
    P.bc_dtype = _lib.dtype_id(Bk if var_b else (Ck if var_c else u))
    for name, v in (("D", D), ("delta_bias", delta_bias)):
        if v is not None:
            if v.dtype != torch.float32 or v.shape != (dim,):
                raise RuntimeError(f"{name} must be float32 (dim,)")
            if v.stride(0) != 1:
                v = v.contiguous()
            setattr(P, name, _lib.ptr(v))
            if name == "D":
                D = v
            else:
                delta_bias = v
    if z is not None:
        if z.shape != u.shape or z.dtype != u.dtype:
            raise RuntimeError("z must match u")
        if out_z is None:
            out_z = torch.empty_like(z)
    if out is None and (want_out or z is None):
        out = torch.empty_like(delta if delta is not None else u)
    for name, t in (("u", u), ("delta", delta), ("z", z), ("out", out), ("out_z", out_z)):
        if t is None:
            continue
        if t.shape != u.shape or t.dtype != u.dtype:
            raise RuntimeError(f"{name} must match u in shape and dtype")
        sb, sd, sl = t.stride()
        setattr(P, name, _lib.ptr(t))
        setattr(P, f"{name}_batch_stride", sb), setattr(P, f"{name}_d_stride", sd), setattr(P, f"{name}_l_stride", sl)
    P.A = _lib.ptr(A)
    P.A_d_stride, P.A_dstate_stride = A.stride()
    if x is not None:
        n_chunks = (L + int(chunk_len) - 1) // int(chunk_len)
        if x.shape != (batch, dim, n_chunks, 2 * N) or x.dtype != torch.float32 or not x.is_contiguous():
            raise RuntimeError("x must be contiguous float32 (batch, dim, n_chunks, 2*dstate)")
        P.x = _lib.ptr(x)
    for name, idx in (("z_row_index", z_row_index), ("out_row_index", out_row_index)):
        if idx is not None:
            if idx.dtype != torch.int32 or idx.shape != (L,) or not idx.is_contiguous():
                raise RuntimeError(f"{name} must be a contiguous int32 tensor of length seqlen")
            setattr(P, name, _lib.ptr(idx))
    if checkpoints is not None:
        if checkpoints.dtype != torch.float32 or not checkpoints.is_contiguous():
            raise RuntimeError("checkpoints must be a contiguous float32 buffer")
        P.checkpoints = _lib.ptr(checkpoints)
    if dt_x is not None:
        if (dt_x.dim() != 3 or dt_x.shape[0] != batch or dt_x.shape[1] != L or dt_x.dtype != u.dtype or dt_x.stride(2) != 1
                or dt_w.dim() != 2 or dt_w.shape[0] != dim or dt_w.dtype != u.dtype or dt_w.stride(1) != 1 or dt_x.shape[2] < dt_w.shape[1]):
            raise RuntimeError("dt_x must be (batch, seqlen, >= dt_rank) rows and dt_w (dim, dt_rank), both in the dtype of u")
        P.dt_x, P.dt_w, P.dt_rank = _lib.ptr(dt_x), _lib.ptr(dt_w), dt_w.shape[1]
        P.dt_x_batch_stride, P.dt_x_l_stride, P.dt_w_row_stride = dt_x.stride(0), dt_x.stride(1), dt_w.stride(0)
    P.reset_period = int(reset_period)       # > 0: independent sequences of that many steps along seqlen
    status = (_lib.C.c_int32 * 2)(0, 0)
    P.info = status                          # host out-field: which kernel family ran, whether it writes the checkpoints
    _lib.call("zigma_selective_scan_fwd", P, dev)
    if info is not None:
        info[:] = [int(status[0]), int(status[1])]
    return out, out_z


def x_proj_eligible(u, weight):
    """limits of zigma_x_proj_fwd: bf16, n <= 96, k % 256 == 0, 16-byte aligned contiguous rows.  From 16 384 tokens on a workgroup streams
    256 token rows over the whole K; below, K is split over the waves of 32-token workgroups (x_proj_splitk_kernel, round 5: the streaming
    form was 32 workgroups and 31 us at 8192 tokens against 21 for the library), which needs k <= 1536."""
    tokens = u.numel() // u.shape[-1]
    return (u.is_cuda and u.dtype == torch.bfloat16 and weight.dtype == torch.bfloat16 and u.is_contiguous()
            and (tokens >= 16384 or (tokens >= 256 and weight.shape[1] <= 1536))
            and weight.shape[0] <= 96 and weight.shape[1] % 256 == 0 and weight.stride(1) == 1 and weight.stride(0) % 8 == 0
            and u.data_ptr() % 16 == 0 and weight.data_ptr() % 16 == 0)


def x_proj(u, weight):
    """x_dbl = u @ weight.T on the matrix cores with a kernel shaped for the skinny output (zigma_x_proj_fwd)."""
    dev = _lib.require_device(u, weight)
    lead, K = u.shape[:-1], u.shape[-1]
    u2 = u.reshape(-1, K)
    n = weight.shape[0]
    out = torch.empty(u2.shape[0], n, device=u.device, dtype=u.dtype)
    P = _lib.XProjParams()
    P.m, P.n, P.k, P.dtype, P.flags = u2.shape[0], n, K, _lib.dtype_id(u), 0
    P.x_row_stride, P.w_row_stride, P.out_row_stride = u2.stride(0), weight.stride(0), out.stride(0)
    P.x, P.w, P.out = _lib.ptr(u2), _lib.ptr(weight), _lib.ptr(out)
    _lib.call("zigma_x_proj_fwd", P, dev)
    return out.reshape(*lead, n)


def conv_x_proj_eligible(x_half, conv_w, conv_b, x_proj_weight, perm, reset_period=0):
    """limits of zigma_conv_x_proj_fwd: bf16, width-4 taps as contiguous (d_inner, 4), a bias, seqlen % 32 == 0,
    batch * seqlen % 256 == 0 and >= 16384 positions (a workgroup walks 256 positions over the whole d_inner: fewer than ~64
    workgroups leave the chip idle), d_inner % 64 == 0, n <= 96, 16-byte aligned rows, one sequence per batch row."""
    if conv_b is None or reset_period or not x_half.is_cuda:
        return False
    Bsz, L, Di = x_half.shape
    return (x_half.dtype == torch.bfloat16 and conv_w.dtype == torch.bfloat16 and conv_b.dtype == torch.bfloat16
            and x_proj_weight.dtype == torch.bfloat16 and conv_w.shape == (Di, 4) and conv_w.is_contiguous() and conv_b.is_contiguous()
            and L % 32 == 0 and (Bsz * L) % 256 == 0 and Bsz * L >= CONV_X_PROJ_MIN_POSITIONS and Di % 64 == 0
            and x_proj_weight.shape[0] <= 96 and x_proj_weight.shape[0] % 8 == 0
            and x_half.stride(2) == 1 and x_half.stride(1) % 8 == 0 and x_half.stride(0) % 8 == 0
            and x_proj_weight.stride(1) == 1 and x_proj_weight.stride(0) % 8 == 0
            and all(t.data_ptr() % 16 == 0 for t in (x_half, conv_w, conv_b, x_proj_weight))
            and (perm is None or (perm.dtype == torch.int32 and perm.is_contiguous())))


def conv_x_proj(x_half, conv_w, conv_b, x_proj_weight, perm=None, _flags=0):
    """u = silu(causal_conv1d(x_half[:, perm])) and x_dbl = u @ x_proj_weight.T in one pass over x (zigma_conv_x_proj_fwd).
    x_half: (B, L, d_inner) bf16 view with contiguous channels (the first half of the in_proj output, as is); conv_w: (d_inner, 4);
    returns u (B, L, d_inner) in SCAN order and x_dbl (B, L, n).  Replaces causal_conv1d_fn + F.linear of reference
    selective_scan_interface.py:307-322."""
    dev = _lib.require_device(x_half, conv_w, conv_b, x_proj_weight, perm)
    Bsz, L, Di = x_half.shape
    n = x_proj_weight.shape[0]
    u = torch.empty(Bsz, L, Di, device=x_half.device, dtype=x_half.dtype)
    x_dbl = torch.empty(Bsz, L, n, device=x_half.device, dtype=x_half.dtype)
    P = _lib.ConvXProjParams()
    P.batch, P.seqlen, P.dim, P.n, P.dtype, P.flags = Bsz, L, Di, n, _lib.dtype_id(x_half), _flags
    P.x_batch_stride, P.x_l_stride = x_half.stride(0), x_half.stride(1)
    P.u_batch_stride, P.u_l_stride = u.stride(0), u.stride(1)
    P.w_row_stride, P.out_row_stride = x_proj_weight.stride(0), n
    P.x, P.conv_weight, P.conv_bias, P.w = _lib.ptr(x_half), _lib.ptr(conv_w), _lib.ptr(conv_b), _lib.ptr(x_proj_weight)
    P.u, P.out, P.x_row_index = _lib.ptr(u), _lib.ptr(x_dbl), _lib.ptr(perm)
    _lib.call("zigma_conv_x_proj_fwd", P, dev)
    return u, x_dbl


ACCUMULATE_IN_SCAN = True     # `v2`: the second sweep's add in its scan epilogue (False: the in-place add; A/B in tests / tools)
DT_PROJ_IN_SPLIT = True       # sequence-split mode: dt_proj + softplus inside the split's first pass, which writes delta for the second (round 6)
DT_PROJ_IN_SCAN = True     # dt_proj + softplus in the scan's tile prologue (MFMA) instead of a kernel of its own


from . import _knobs  # noqa: E402
_knobs.apply(globals(), "selective_scan_interface")      # (SPLIT_MAX_WGS, CONV_X_PROJ_MIN_POSITIONS, ... for the A/B tools)


def dt_in_scan_eligible(u, x_dbl, weight, reset_period=0, out=None, dstate=16, z=None):
    """limits of the in-kernel dt_proj of scan_tok2_kernel (zigma_scan_params_t.dt_x), mirroring tok2_dtp_ok() / tok2_layout_ok() /
    tok_eligible() of csrc/: bf16 / fp16, whole-sequence mode of the hot kernel (dstate == 16, seqlen % 16 == 0, d_inner % 64 == 0; reset_period a
    multiple of 16 — the video temporal layers, round 6), 32 <= dt_rank <= 64 and % 8 == 0, x_dbl rows >= 64 wide on 16-byte boundaries, channel-contiguous u / z, and every
    in-sample offset (seqlen * row stride, in bytes of up to 4-byte elements) below 2^31 / 4.  A caller that gets False keeps the
    dt_proj kernel (or F.linear) + the ordinary scan call, which serves every shape the reference does."""
    R = weight.shape[1]
    if not (u.is_cuda and u.dtype in (torch.bfloat16, torch.float16) and x_dbl.dtype == u.dtype and weight.dtype == u.dtype and reset_period >= 0 and reset_period % 16 == 0
            and dstate == 16 and u.dim() == 3 and x_dbl.dim() == 3
            and 32 <= R <= 64 and R % 8 == 0 and u.shape[1] % 16 == 0 and u.shape[2] % 64 == 0 and x_dbl.shape[2] >= 64 and x_dbl.shape[2] >= R + 2 * dstate
            and u.stride(2) == 1 and x_dbl.stride(2) == 1 and x_dbl.stride(1) % 8 == 0 and x_dbl.stride(0) % 8 == 0 and x_dbl.stride(1) >= 64
            and weight.stride(1) == 1 and weight.stride(0) % 8 == 0
            and x_dbl.data_ptr() % 16 == 0 and weight.data_ptr() % 16 == 0 and u.shape[0] <= 65535):
        return False
    L, lim = u.shape[1], ((1 << 31) - 1) // 4
    rows = [u.stride(1), x_dbl.stride(1)] + ([z.stride(1)] if z is not None else []) + ([out.stride(1)] if out is not None else [])
    if z is not None and z.stride(2) != 1:
        return False
    return all(0 <= s and s * L <= lim for s in rows)


def dt_proj_eligible(x_dbl, dt_rank, weight):
    """bf16 token-major x_dbl rows / weight rows on 16-byte boundaries, d_inner a multiple of 64, dt_rank <= 48, % 8 == 0."""
    return (x_dbl.is_cuda and x_dbl.dtype == torch.bfloat16 and weight.dtype == torch.bfloat16 and dt_rank <= 48
            and dt_rank % 8 == 0
            and weight.shape[0] % 64 == 0 and x_dbl.stride(-1) == 1 and weight.stride(1) == 1
            and x_dbl.stride(-2) % 8 == 0 and weight.stride(0) % 8 == 0
            and x_dbl.data_ptr() % 16 == 0 and weight.data_ptr() % 16 == 0)


def dt_proj_softplus(x_dbl, dt_rank, weight, bias=None, softplus=True):
    """delta' = softplus(x_dbl[..., :dt_rank] @ weight.T + bias) on the matrix cores (zigma_dt_proj_softplus_fwd).
    x_dbl: (..., >= dt_rank) bf16 rows; weight: (d_inner, dt_rank) bf16; bias: float32 (d_inner) or None.
    Returns (..., d_inner) bf16.  Fuses reference selective_scan_interface.py:323 with the softplus(delta + bias)
    at the top of its scan kernel; the scan is then called with delta_softplus=False, delta_bias=None."""
    dev = _lib.require_device(x_dbl, weight, bias)
    lead = x_dbl.shape[:-1]
    x2 = x_dbl.reshape(-1, x_dbl.shape[-1])
    if x2.stride(-1) != 1:
        raise RuntimeError("x_dbl rows must be contiguous")
    n = weight.shape[0]
    out = torch.empty(x2.shape[0], n, device=x_dbl.device, dtype=x_dbl.dtype)
    P = _lib.DtProjParams()
    P.m, P.n, P.k = x2.shape[0], n, dt_rank
    P.dtype, P.softplus, P.flags = _lib.dtype_id(x_dbl), int(bool(softplus)), DT_PROJ_FLAGS
    P.x_row_stride, P.w_row_stride, P.out_row_stride = x2.stride(0), weight.stride(0), out.stride(0)
    P.x, P.w, P.out = _lib.ptr(x2), _lib.ptr(weight), _lib.ptr(out)
    if bias is not None:
        if bias.dtype != torch.float32 or bias.shape != (n,) or bias.stride(0) != 1:
            raise RuntimeError("bias must be contiguous float32 (d_inner,)")
        P.bias = _lib.ptr(bias)
    _lib.call("zigma_dt_proj_softplus_fwd", P, dev)
    return out.reshape(*lead, n)


BWD_MAX_BATCH = 65535        # batch limit of one zigma_selective_scan_bwd / zigma_causal_conv1d_bwd launch (grid dimension)


def scan_bwd_tok(u, delta, A, B, C, D, z, delta_bias, dout, out, delta_softplus, *, dB=None, dC=None, dz=None,
                 z_row_index=None, out_row_index=None, checkpoints=None, reset_period=0):
    """Backward of the token-major selective scan (zigma_selective_scan_bwd; reference selective_scan_cuda.bwd,
    selective_scan.cpp:338-492).

    u, delta, z, out, dout: (batch, seqlen, dim), channel stride 1; out = the forward's UNGATED y (only with z).
    B, C: (batch, seqlen, dstate)-shaped views (any strides), same dtype.  A (dim, dstate) f32, D / delta_bias f32.
    dB, dC: optional preallocated float32 (batch, seqlen, dstate) views to write into (e.g. columns of d(x_dbl)).
    dz: optional preallocated (batch, seqlen, dim) view (e.g. the z half of d(xz)).
    z_row_index / out_row_index: the forward's int32 row tables — z / dz live at row z_row_index[k], out / dout at row
    out_row_index[k] of their tensors (token order) while u, delta, B, C, du, ddelta, dB, dC are in scan order.
    Returns du, ddelta, dA, dB, dC, dD, dz, ddelta_bias (None where the input was None); du/ddelta/dz in the input dtype,
    the rest float32."""
    dev = _lib.require_device(u, delta, A, B, C, D, z, delta_bias, dout, out)
    Bsz, L, Dm = u.shape
    N = A.shape[1]
    if Bsz > BWD_MAX_BATCH:
        # the kernel rides the batch in a 16-bit grid dimension: larger batches (the differentiable video temporal path reshapes to
        # batch * tokens-per-frame rows) go in slices; per-sample outputs land in views of one allocation, the parameter gradients
        # (dA, dD, ddelta_bias) are summed over the slices
        du, ddelta = torch.empty_like(u, memory_format=torch.contiguous_format), torch.empty_like(u, memory_format=torch.contiguous_format)
        if z is not None and dz is None:
            dz = torch.empty_like(u, memory_format=torch.contiguous_format)
        f32 = dict(device=u.device, dtype=torch.float32)
        dB = torch.empty(Bsz, L, N, **f32) if dB is None else dB
        dC = torch.empty(Bsz, L, N, **f32) if dC is None else dC
        dA = dD = dbias = None
        sl = lambda t, a, b: None if t is None else t[a:b]
        for a in range(0, Bsz, BWD_MAX_BATCH):
            b = min(a + BWD_MAX_BATCH, Bsz)
            r = scan_bwd_tok(u[a:b], delta[a:b], A, B[a:b], C[a:b], D, sl(z, a, b), delta_bias, dout[a:b], sl(out, a, b), delta_softplus,
                             dB=dB[a:b], dC=dC[a:b], dz=sl(dz, a, b), z_row_index=z_row_index, out_row_index=out_row_index,
                             checkpoints=None if checkpoints is None else checkpoints[a:b], reset_period=reset_period)
            du[a:b].copy_(r[0]); ddelta[a:b].copy_(r[1])
            dA = r[2] if dA is None else dA + r[2]
            dD = r[5] if dD is None or r[5] is None else dD + r[5]
            dbias = r[7] if dbias is None or r[7] is None else dbias + r[7]
        return du, ddelta, dA, dB, dC, dD, dz, dbias
    for t, nm in ((u, "u"), (delta, "delta"), (z, "z"), (out, "out"), (dout, "dout")):
        if t is not None and (t.shape != (Bsz, L, Dm) or t.stride(2) != 1 or t.dtype != u.dtype):
            raise RuntimeError(f"{nm} must be (batch, seqlen, dim) with channel stride 1 and the dtype of u")
    if B.shape != (Bsz, L, N) or C.shape != (Bsz, L, N) or B.dtype != u.dtype or C.dtype != u.dtype:
        raise RuntimeError("B, C must be (batch, seqlen, dstate) views in the dtype of u")
    if A.dtype != torch.float32:
        raise RuntimeError("A must be float32")
    if z is not None and out is None:
        raise RuntimeError("the gated backward needs the forward's ungated `out`")
    du, ddelta = torch.empty_like(u, memory_format=torch.contiguous_format), torch.empty_like(u, memory_format=torch.contiguous_format)
    if z is not None and dz is None:
        dz = torch.empty_like(u, memory_format=torch.contiguous_format)
    if dz is not None and (dz.shape != u.shape or dz.stride(2) != 1 or dz.dtype != u.dtype):
        raise RuntimeError("dz must be (batch, seqlen, dim) with channel stride 1 and the dtype of u")
    f32 = dict(device=u.device, dtype=torch.float32)
    dA = torch.zeros(Dm, N, **f32)
    dB = torch.empty(Bsz, L, N, **f32) if dB is None else dB
    dC = torch.empty(Bsz, L, N, **f32) if dC is None else dC
    dD = torch.zeros(Dm, **f32) if D is not None else None
    dbias = torch.zeros(Dm, **f32) if delta_bias is not None else None
    P = _lib.ScanBwdParams()
    P.batch, P.dim, P.seqlen, P.dstate = Bsz, Dm, L, N
    P.delta_softplus, P.io_dtype, P.flags = int(bool(delta_softplus)), _lib.dtype_id(u), 0
    P.reset_period = int(reset_period)       # > 0: independent sequences of that many steps along seqlen (as in the forward)
    for name, t in (("u", u), ("delta", delta), ("z", z), ("out", out), ("dout", dout), ("du", du), ("ddelta", ddelta),
                    ("dz", dz)):
        if t is not None:
            setattr(P, name, _lib.ptr(t))
            setattr(P, name + "_batch_stride", t.stride(0))
            setattr(P, name + "_l_stride", t.stride(1))
    P.A, P.A_d_stride, P.A_dstate_stride = _lib.ptr(A), A.stride(0), A.stride(1)
    for name, t in (("B", B), ("C", C), ("dB", dB), ("dC", dC)):
        setattr(P, name, _lib.ptr(t))
        setattr(P, name + "_batch_stride", t.stride(0))
        setattr(P, name + "_l_stride", t.stride(1))
        setattr(P, name + "_dstate_stride", t.stride(2))
    for t, nm in ((D, "D"), (delta_bias, "delta_bias")):
        if t is not None and (t.dtype != torch.float32 or not t.is_contiguous()):
            raise RuntimeError(f"{nm} must be contiguous float32")
    P.D, P.delta_bias, P.dA, P.dD, P.ddelta_bias = _lib.ptr(D), _lib.ptr(delta_bias), _lib.ptr(dA), _lib.ptr(dD), _lib.ptr(dbias)
    for name, tab in (("z_row_index", z_row_index), ("out_row_index", out_row_index)):
        if tab is not None:
            if tab.dtype != torch.int32 or tab.shape != (L,) or not tab.is_contiguous():
                raise RuntimeError(f"{name} must be a contiguous int32 (seqlen,) table")
            setattr(P, name, _lib.ptr(tab))
    if checkpoints is not None:            # written by the forward kernel (scan_raw(..., checkpoints=)): skip phase 1
        if checkpoints.dtype != torch.float32 or not checkpoints.is_contiguous() or \
                checkpoints.numel() != Bsz * (Dm // 64) * ((L + 15) // 16) * N * 64:
            raise RuntimeError("checkpoints: float32 [batch][dim/64][ceil(seqlen/16)][dstate][64]")
        P.checkpoints = _lib.ptr(checkpoints)
    ws = _lib.workspace("zigma_selective_scan_bwd", P, dev)
    _lib.call("zigma_selective_scan_bwd", P, dev)
    del ws
    return du, ddelta, dA, dB, dC, dD, dz, dbias


class MambaInnerTokFn(torch.autograd.Function):
    """Autograd form of the token-major Mamba inner (conv + SiLU -> x_proj -> dt_proj -> gated selective scan) with the
    zigzag reordering fused into the kernels' row tables in BOTH directions (no index_select / index_add / cat):
    forward  = conv (x_row_index) -> GEMMs -> scan (z_row_index, out_row_index), saving u, x_dbl, delta and the ungated y;
    backward = scan bwd (same tables; dz lands in token order) -> the two skinny GEMM pairs -> conv bwd (dx scattered
    through x_row_index).  Mirrors MambaInnerFn (selective_scan_interface.py:296-434) without its out_proj."""

    @staticmethod
    def forward(ctx, xz, conv_w, conv_b, x_proj_w, dt_proj_w, A, D, delta_bias, perm, out_rows, reset_period=0):
        # reset_period > 0 (the video temporal layers): xz is the (k, b * t, 2 Di) strided VIEW of the (b * t, k, 2 Di) projection output —
        # batch = pixel, sequence = the frames of every sample one after the other, conv window and state restart every t steps; the
        # result and d(xz) are views of (b * t, k, .) allocations, so neither direction pays a transposing copy
        Bsz, L, C2 = xz.shape
        Di, R, N = C2 // 2, dt_proj_w.shape[1], A.shape[1]
        w = conv_w.reshape(Di, -1)
        x_half, z_half = xz[:, :, :Di], xz[:, :, Di:]
        if USE_CONV_X_PROJ and conv_x_proj_eligible(x_half, w, conv_b, x_proj_w, perm, reset_period):
            u, x_dbl = conv_x_proj(x_half, w, conv_b, x_proj_w, perm)      # one pass; the backward takes u and x_dbl as saved
        else:
            u = torch.empty(Bsz, L, Di, device=xz.device, dtype=xz.dtype)
            causal_conv1d_raw(x_half.transpose(1, 2), w, conv_b, True, out=u.transpose(1, 2), x_row_index=perm, reset_period=reset_period)
            x_dbl = F.linear(u, x_proj_w)
        delta = F.linear(x_dbl[:, :, :R], dt_proj_w)
        Bm, Cm = x_dbl[:, :, R:R + N], x_dbl[:, :, R + N:R + 2 * N]
        strided = not xz.is_contiguous()
        out = torch.empty(Bsz, L, Di, device=xz.device, dtype=xz.dtype)
        # (strided input = a transposed view: hand the result back as the same kind of view of an (L, Bsz, Di) allocation)
        y = torch.empty(L, Bsz, Di, device=xz.device, dtype=xz.dtype).transpose(0, 1) if strided else torch.empty(Bsz, L, Di, device=xz.device, dtype=xz.dtype)
        # the states before every 16-step tile, for the backward's reverse sweep (the token-major kernel writes them on
        # its way: 4 * Di * N * L / 16 bytes per sample; anything else leaves the buffer alone and the backward recomputes)
        ck = None
        if Di % 64 == 0 and N in (8, 16):
            ck = torch.empty(Bsz, Di // 64, (L + 15) // 16, N, 64, device=xz.device, dtype=torch.float32)
        info = []
        scan_raw(u.transpose(1, 2), delta.transpose(1, 2), A, Bm.transpose(1, 2).unsqueeze(1),
                 Cm.transpose(1, 2).unsqueeze(1), D, z_half.transpose(1, 2), delta_bias, True,
                 out=out.transpose(1, 2), out_z=y.transpose(1, 2), z_row_index=perm, out_row_index=out_rows, checkpoints=ck,
                 info=info, reset_period=reset_period)
        if ck is not None and info[1] != 1:          # the kernel that served the call does not write checkpoints
            ck = None
        ctx.save_for_backward(xz, conv_w, conv_b, x_proj_w, dt_proj_w, A, D, delta_bias, u, x_dbl, delta, out)
        ctx.perm, ctx.out_rows, ctx.ck, ctx.reset_period = perm, out_rows, ck, reset_period
        return y

    @staticmethod
    def backward(ctx, dy):
        xz, conv_w, conv_b, x_proj_w, dt_proj_w, A, D, delta_bias, u, x_dbl, delta, out = ctx.saved_tensors
        Bsz, L, C2 = xz.shape
        Di, R, N = C2 // 2, dt_proj_w.shape[1], A.shape[1]
        x_half, z_half = xz[:, :, :Di], xz[:, :, Di:]
        if dy.stride(2) != 1:
            dy = dy.contiguous()
        rp = ctx.reset_period
        dxz = torch.empty(L, Bsz, C2, device=xz.device, dtype=xz.dtype).transpose(0, 1) if not xz.is_contiguous() else torch.empty_like(xz)
        dx_dbl = torch.empty(Bsz, L, R + 2 * N, device=xz.device, dtype=torch.float32)
        du, ddelta, dA, _, _, dD, _, dbias = scan_bwd_tok(
            u, delta, A, x_dbl[:, :, R:R + N], x_dbl[:, :, R + N:R + 2 * N], D, z_half, delta_bias, dy, out, True,
            dB=dx_dbl[:, :, R:R + N], dC=dx_dbl[:, :, R + N:], dz=dxz[:, :, Di:], z_row_index=ctx.perm,
            out_row_index=ctx.out_rows, checkpoints=ctx.ck, reset_period=rp)
        ctx.ck = None
        dd2 = ddelta.reshape(-1, Di)
        dx_dbl[:, :, :R] = (dd2 @ dt_proj_w).reshape(Bsz, L, R)                  # d(x_dbl[:, :R]) = ddelta @ W_dt
        d_dt_w = wgrad(dd2, x_dbl.reshape(-1, R + 2 * N)[:, :R].contiguous())       # (Di, R)  token-slab GEMMs (zigma_amd/wgrad.py)
        dxd = dx_dbl.to(xz.dtype).reshape(-1, R + 2 * N)
        du = torch.addmm(du.reshape(-1, Di), dxd, x_proj_w).reshape(Bsz, L, Di)    # x_dbl = u @ W_x^T
        d_x_w = wgrad(dxd, u.reshape(-1, Di))                                      # (R + 2N, Di)
        _, d_cw, d_cb = conv_bwd_tok(x_half, conv_w, conv_b, du, True, ctx.perm, dx=dxz[:, :, :Di], reset_period=rp)
        return (dxz, d_cw.to(conv_w.dtype).reshape(conv_w.shape), None if d_cb is None else d_cb.to(conv_b.dtype),
                d_x_w.to(x_proj_w.dtype), d_dt_w.to(dt_proj_w.dtype), dA.to(A.dtype), dD.to(D.dtype),
                None if dbias is None else dbias.to(delta_bias.dtype), None, None, None)


def mamba_inner_tok_train(xz, conv1d_weight, conv1d_bias, x_proj_weight, delta_proj_weight, A, D, delta_bias, *,
                          perm=None, out_rows=None, reset_period=0):
    """Differentiable mamba_inner_tok (same row-table and reset_period semantics; a strided xz view gets a strided result view)."""
    if D is None or delta_bias is None:
        raise RuntimeError("the differentiable path expects D and delta_bias (ZigMa always has them)")
    return MambaInnerTokFn.apply(xz, conv1d_weight, conv1d_bias, x_proj_weight, delta_proj_weight, A, D, delta_bias, perm,
                                 perm if out_rows is None else out_rows, int(reset_period))


def selective_scan_cuda_fwd(u, delta, A, B, C, D_, z_, delta_bias_, delta_softplus):
    """Drop-in for the extension entry `selective_scan_cuda.fwd` -> [out, x] (+ [out_z] if z).
    Callee allocates: out = empty_like(delta), x (B, D, ceil(L/2048), 2N) f32, out_z = empty_like(z)."""
    batch, dim, L = u.shape
    x = torch.empty(batch, dim, (L + 2047) // 2048, 2 * A.shape[1], device=u.device, dtype=torch.float32)
    out, out_z = scan_raw(u, delta, A, B, C, D_, z_, delta_bias_, delta_softplus, x=x)
    return [out, x] if z_ is None else [out, x, out_z]


def selective_scan_fn(u, delta, A, B, C, D=None, z=None, delta_bias=None, delta_softplus=False,
                      return_last_state=False):
    """if return_last_state is True, returns (out, last_state); last_state is (batch, dim, dstate)."""
    batch, dim, L = u.shape
    x = None
    if return_last_state:
        x = torch.empty(batch, dim, (L + 2047) // 2048, 2 * A.shape[1], device=u.device, dtype=torch.float32)
    out, out_z = scan_raw(u, delta, A, B, C, D, z, delta_bias, delta_softplus, x=x, want_out=z is None)
    res = out if z is None else out_z
    return (res, x[:, :, -1, 1::2]) if return_last_state else res


def _inner_common(xz, conv1d_weight, conv1d_bias, x_proj_weight, delta_proj_weight, A, B, C, D, delta_bias,
                  B_proj_bias, C_proj_bias, delta_softplus):
    """Reference-layout inner: xz (batch, 2*d_inner, seqlen).  Runs on the token-major kernels internally
    (one transposing copy in, none out)."""
    if B is not None or C is not None:
        raise RuntimeError("zigma_amd: only input-dependent B and C are supported (ZigMa always passes None)")
    xz_tok = xz.transpose(1, 2).contiguous()                      # (B, L, 2Di)
    return mamba_inner_tok(xz_tok, conv1d_weight, conv1d_bias, x_proj_weight, delta_proj_weight, A, D, delta_bias,
                           B_proj_bias=B_proj_bias, C_proj_bias=C_proj_bias, delta_softplus=delta_softplus)


def mamba_inner_fn(xz, conv1d_weight, conv1d_bias, x_proj_weight, delta_proj_weight, out_proj_weight, out_proj_bias,
                   A, B=None, C=None, D=None, delta_bias=None, B_proj_bias=None, C_proj_bias=None,
                   delta_softplus=True):
    """xz: (batch, 2*dim, seqlen) -> (batch, seqlen, d_model)."""
    y = _inner_common(xz, conv1d_weight, conv1d_bias, x_proj_weight, delta_proj_weight, A, B, C, D, delta_bias,
                      B_proj_bias, C_proj_bias, delta_softplus)
    return F.linear(y, out_proj_weight, out_proj_bias)


def mamba_inner_fn_no_out_proj(xz, conv1d_weight, conv1d_bias, x_proj_weight, delta_proj_weight, A, B=None, C=None,
                               D=None, delta_bias=None, B_proj_bias=None, C_proj_bias=None, delta_softplus=True):
    """xz: (batch, 2*dim, seqlen) -> out_z (batch, dim, seqlen)."""
    y = _inner_common(xz, conv1d_weight, conv1d_bias, x_proj_weight, delta_proj_weight, A, B, C, D, delta_bias,
                      B_proj_bias, C_proj_bias, delta_softplus)
    return y.transpose(1, 2)


def mamba_inner_tok(xz, conv1d_weight, conv1d_bias, x_proj_weight, delta_proj_weight, A, D, delta_bias, *,
                    perm=None, out_rows=None, B_proj_bias=None, C_proj_bias=None, delta_softplus=True, out=None,
                    reset_period=0, z_preactivated=False, add_to=None):
    """Token-major Mamba inner (no out_proj).

    xz: (batch, seqlen, 2*d_inner), token order, channel contiguous (the in_proj GEMM output as is).
    perm: optional int32 (seqlen,) device table; the recurrence runs over tokens perm[0], perm[1], ...
          (zigzag / Hilbert order, reversed order for the backward sweep of `v2`); the result comes back
          in TOKEN order.  Semantics of mamba_simple.py:362-395: x'[k] = x[perm[k]], out[perm[k]] = out'[k].
    out_rows: optional int32 (seqlen,) table for the write-back when it is NOT the inverse of the gather:
          out[out_rows[k]] = out'[k] (defaults to perm).  The reference pairs its temporal table [0..T-1] with
          the "reverse" table [T-1..0] (model_zigma.py:765-772), i.e. out = out'[:, perm_rev] with perm_rev not
          the inverse of perm; the caller passes out_rows = inverse(perm_rev) to reproduce exactly that.
    reset_period: > 0 = every batch row is a concatenation of independent sequences of that many steps (multiple of 16):
          conv window and SSM state restart there (the video temporal layers: batch = k, seqlen = b * t on strided views).
    z_preactivated: the z half of xz already holds silu(z) (an in_proj epilogue wrote it; inference, 16-bit, d_state 16,
          seqlen % 16 == 0 only — the hot kernel's ZIGMA_SCAN_Z_PREACTIVATED form).
    add_to: optional (batch, seqlen, d_inner) tensor y0 in token order (inference only): the result is y0 + y, written INTO y0 and returned — the second
          sweep of `v2` (mamba_simple.py:335-339).  Where the in-kernel dt_proj form of the scan serves the call the add rides in its epilogue
          (ZIGMA_SCAN_ACCUMULATE: no elementwise pass); otherwise it is an in-place add behind the scan.
    Returns y (batch, seqlen, d_inner) in token order = out_z of the reference's scan, before out_proj.
    """
    if xz.dim() != 3 or xz.stride(2) != 1:
        raise RuntimeError("xz must be (batch, seqlen, 2*d_inner) with contiguous channels")
    if torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in (
            xz, conv1d_weight, conv1d_bias, x_proj_weight, delta_proj_weight, A, D, delta_bias)):
        if B_proj_bias is not None or C_proj_bias is not None or not delta_softplus or out is not None or z_preactivated or add_to is not None:
            raise NotImplementedError("differentiable mamba_inner_tok: no B/C projection bias, softplus on, no out= / add_to=, no pre-activated gate")
        return mamba_inner_tok_train(xz, conv1d_weight, conv1d_bias, x_proj_weight, delta_proj_weight, A, D, delta_bias,
                                     perm=perm, out_rows=out_rows, reset_period=reset_period)
    Bsz, L, C2 = xz.shape
    Di = C2 // 2
    R = delta_proj_weight.shape[1]
    N = A.shape[1]
    w = conv1d_weight.reshape(Di, -1)
    x_half, z_half = xz[:, :, :Di], xz[:, :, Di:]
    if USE_CONV_X_PROJ and conv_x_proj_eligible(x_half, w, conv1d_bias, x_proj_weight, perm, reset_period):
        u, x_dbl = conv_x_proj(x_half, w, conv1d_bias, x_proj_weight, perm)   # one pass: read x, write u (scan order) and x_dbl
    else:
        # depthwise causal conv + SiLU over the reordered sequence; u is in SCAN order
        u = torch.empty(Bsz, L, Di, device=xz.device, dtype=xz.dtype)
        causal_conv1d_raw(x_half.transpose(1, 2), w, conv1d_bias, True, out=u.transpose(1, 2), x_row_index=perm,
                          reset_period=reset_period)
        if USE_X_PROJ_KERNEL and x_proj_eligible(u, x_proj_weight):
            x_dbl = x_proj(u, x_proj_weight)                             # (B, L, R + 2N)   read-bound MFMA kernel
        else:
            x_dbl = F.linear(u, x_proj_weight)                           # (B, L, R + 2N)   GEMM
    return _inner_tok_tail(u, x_dbl, z_half, delta_proj_weight, A, D, delta_bias, perm, out_rows, B_proj_bias, C_proj_bias,
                           delta_softplus, out, reset_period, z_preactivated, add_to)


def _inner_tok_tail(u, x_dbl, z_half, delta_proj_weight, A, D, delta_bias, perm, out_rows, B_proj_bias, C_proj_bias,
                    delta_softplus, out, reset_period, z_preactivated, add_to=None):
    """dt_proj (+ softplus) and the scan over u (scan order), x_dbl, z (token order): the part of the inner function behind x_proj"""
    Bsz, L, Di = u.shape
    R = delta_proj_weight.shape[1]
    N = A.shape[1]
    if add_to is not None and out is not None:
        raise RuntimeError("mamba_inner_tok: pass out= or add_to=, not both")
    in_scan = (DT_PROJ_IN_SCAN and delta_softplus
               and dt_in_scan_eligible(u, x_dbl, delta_proj_weight, reset_period, out if add_to is None else add_to, dstate=N, z=z_half)
               and B_proj_bias is None and C_proj_bias is None and not split_chunk_len(Bsz, Di, L, reset_period))
    chunk_len_split = split_chunk_len(Bsz, Di, L, reset_period)
    in_split = (DT_PROJ_IN_SPLIT and not in_scan and chunk_len_split and delta_softplus and B_proj_bias is None and C_proj_bias is None and delta_bias is not None
                and not z_preactivated and u.dtype == z_half.dtype and dt_in_scan_eligible(u, x_dbl, delta_proj_weight, reset_period, out if add_to is None else add_to, dstate=N, z=z_half)
                and Bsz * (Di // 64) < 768 and chunk_len_split % 16 == 0 and -(-L // chunk_len_split) >= 2)
    if in_scan:                      # dt_proj + bias + softplus inside the scan kernel's tile prologue: delta is never materialised
        delta = None
    elif in_split:                   # sequence split: the first pass forms delta and writes it into this workspace for the second (no dt_proj kernel)
        delta = torch.empty(Bsz, L, Di, device=u.device, dtype=u.dtype)
    elif delta_softplus and dt_proj_eligible(x_dbl, R, delta_proj_weight):   # K = dt_rank GEMM + bias + softplus in one write-bound MFMA kernel; scan skips its softplus
        delta = dt_proj_softplus(x_dbl, R, delta_proj_weight, delta_bias, True)
        delta_bias, delta_softplus = None, False
    else:
        delta = F.linear(x_dbl[:, :, :R], delta_proj_weight)         # (B, L, Di)       GEMM
    Bm, Cm = x_dbl[:, :, R:R + N], x_dbl[:, :, R + N:R + 2 * N]
    if B_proj_bias is not None:
        Bm = Bm + B_proj_bias.to(Bm.dtype)
    if C_proj_bias is not None:
        Cm = Cm + C_proj_bias.to(Cm.dtype)
    acc = add_to is not None and in_scan and ACCUMULATE_IN_SCAN          # the scan's epilogue adds to add_to; else an in-place add behind it
    y = add_to if acc else out if out is not None else torch.empty(Bsz, L, Di, device=u.device, dtype=u.dtype)
    # few workgroups (small batch, or long sequences of few samples): hand the kernel a carry buffer and a chunk length so
    # that it splits the sequence over ~768 workgroups (3 per CU): chunk-local states -> combine -> seeded second pass
    xc, chunk_len = None, split_chunk_len(Bsz, Di, L, reset_period)
    if chunk_len:
        xc = torch.empty(Bsz, Di, -(-L // chunk_len), 2 * N, device=u.device, dtype=torch.float32)
    else:
        chunk_len = 2048
    dt_inside = in_scan or in_split
    scan_raw(u.transpose(1, 2), None if in_scan else delta.transpose(1, 2), A, Bm.transpose(1, 2).unsqueeze(1),
             Cm.transpose(1, 2).unsqueeze(1), D, z_half.transpose(1, 2), delta_bias, delta_softplus,
             out_z=y.transpose(1, 2), z_row_index=perm, out_row_index=perm if out_rows is None else out_rows,
             want_out=False, x=xc, reset_period=reset_period, chunk_len=chunk_len, z_preactivated=z_preactivated,
             dt_x=x_dbl if dt_inside else None, dt_w=delta_proj_weight if dt_inside else None, accumulate=acc)
    return y if (add_to is None or acc) else add_to.add_(y)
