"""Host side of zigma_in_conv_x_proj_fwd as it stood in zigma_amd/ (round 4, archived with the kernel): the ctypes block,
the wrappers of selective_scan_interface.py and the Mamba entry that started at the hidden states.  Not importable as is."""

# ---- zigma_amd/_lib.py
class InConvXProjParams(C.Structure):
    _fields_ = ([(n, i32) for n in ("batch", "seqlen", "dim", "n", "k", "dtype", "flags", "pad_")]
                + [(n, i64) for n in ("h_batch_stride", "h_l_stride", "u_batch_stride", "u_l_stride", "win_row_stride", "w_row_stride",
                                      "out_row_stride")]
                + [(n, vp) for n in ("h", "w_in", "conv_weight", "conv_bias", "w", "u", "out", "x_row_index", "workspace")]
                + [("workspace_bytes", i64)])



# ---- zigma_amd/selective_scan_interface.py
def in_conv_x_proj_eligible(h, in_w_x, conv_w, conv_b, x_proj_weight, perm, reset_period=0):
    """limits of zigma_in_conv_x_proj_fwd (the x half of in_proj inside the conv + x_proj kernel): bf16, d_model 640 or 768,
    seqlen % 128 == 0, >= 16384 positions (one workgroup per 128 or 256 positions and CU), d_inner % 64 == 0 and <= 1536,
    width-4 taps as contiguous (d_inner, 4), a bias, n <= 80, 16-byte aligned contiguous rows, one sequence per batch row."""
    if conv_b is None or reset_period or not h.is_cuda or h.dim() != 3:
        return False
    Bsz, L, E = h.shape
    Di = in_w_x.shape[0]
    return (h.dtype == torch.bfloat16 and in_w_x.dtype == torch.bfloat16 and conv_w.dtype == torch.bfloat16
            and conv_b.dtype == torch.bfloat16 and x_proj_weight.dtype == torch.bfloat16
            and E in (640, 768) and in_w_x.shape[1] == E and conv_w.shape == (Di, 4) and conv_w.is_contiguous() and conv_b.is_contiguous()
            and L % 128 == 0 and Bsz * L >= IN_CONV_X_PROJ_MIN_POSITIONS and Di % 64 == 0 and Di <= 1536
            and x_proj_weight.shape[0] <= 80 and x_proj_weight.shape[0] % 8 == 0 and x_proj_weight.shape[1] == Di
            and h.stride(2) == 1 and h.stride(1) % 8 == 0 and h.stride(0) % 8 == 0
            and in_w_x.stride(1) == 1 and in_w_x.stride(0) % 8 == 0
            and x_proj_weight.stride(1) == 1 and x_proj_weight.stride(0) % 8 == 0
            and all(t.data_ptr() % 16 == 0 for t in (h, in_w_x, conv_w, conv_b, x_proj_weight))
            and (perm is None or (perm.dtype == torch.int32 and perm.is_contiguous())))


def in_conv_x_proj(h, in_w_x, conv_w, conv_b, x_proj_weight, perm=None, _flags=0):
    """x = h @ in_w_x.T (the x rows of in_proj.weight), u = silu(causal_conv1d(x[:, perm])) and x_dbl = u @ x_proj_weight.T in
    one kernel (zigma_in_conv_x_proj_fwd): x never reaches memory.  h: (B, L, d_model) bf16 token-major; in_w_x: (d_inner, d_model)
    = in_proj.weight[:d_inner]; returns u (B, L, d_inner) in SCAN order and x_dbl (B, L, n).  Replaces the x columns of the in_proj
    GEMM (reference mamba_simple.py:290-294) + causal_conv1d_fn + F.linear of selective_scan_interface.py:307-322."""
    dev = _lib.require_device(h, in_w_x, conv_w, conv_b, x_proj_weight, perm)
    Bsz, L, E = h.shape
    Di, n = in_w_x.shape[0], x_proj_weight.shape[0]
    u = torch.empty(Bsz, L, Di, device=h.device, dtype=h.dtype)
    x_dbl = torch.empty(Bsz, L, n, device=h.device, dtype=h.dtype)
    P = _lib.InConvXProjParams()
    P.batch, P.seqlen, P.dim, P.n, P.k, P.dtype, P.flags = Bsz, L, Di, n, E, _lib.dtype_id(h), _flags
    P.h_batch_stride, P.h_l_stride = h.stride(0), h.stride(1)
    P.u_batch_stride, P.u_l_stride = u.stride(0), u.stride(1)
    P.win_row_stride, P.w_row_stride, P.out_row_stride = in_w_x.stride(0), x_proj_weight.stride(0), n
    P.h, P.w_in, P.conv_weight, P.conv_bias, P.w = _lib.ptr(h), _lib.ptr(in_w_x), _lib.ptr(conv_w), _lib.ptr(conv_b), _lib.ptr(x_proj_weight)
    P.u, P.out, P.x_row_index = _lib.ptr(u), _lib.ptr(x_dbl), _lib.ptr(perm)
    ws = _lib.workspace("zigma_in_conv_x_proj_fwd", P, dev)      # the x rows in front of every workgroup's first tile (pre-pass)
    _lib.call("zigma_in_conv_x_proj_fwd", P, dev)
    del ws
    return u, x_dbl



Z_HALF_SILU_IN_EPILOGUE = False      # the z projection writes silu(z) and the scan skips its SiLU (ZIGMA_SCAN_Z_PREACTIVATED)


def mamba_inner_hidden_eligible(h, in_proj_weight, conv1d_weight, conv1d_bias, x_proj_weight, perm, reset_period=0):
    """inference path that starts at the block's normalised hidden states: x half of in_proj inside the conv + x_proj kernel"""
    if not USE_IN_CONV_X_PROJ or torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in (
            h, in_proj_weight, conv1d_weight, conv1d_bias, x_proj_weight)):
        return False
    Di = in_proj_weight.shape[0] // 2
    return in_conv_x_proj_eligible(h, in_proj_weight[:Di], conv1d_weight.reshape(Di, -1), conv1d_bias, x_proj_weight, perm, reset_period)


def mamba_inner_hidden(h, in_proj_weight, conv1d_weight, conv1d_bias, x_proj_weight, delta_proj_weight, A, D, delta_bias, *,
                       perm=None, out_rows=None, delta_softplus=True, out=None):
    """Token-major Mamba inner INCLUDING in_proj (no out_proj): h (batch, seqlen, d_model) -> y (batch, seqlen, d_inner).
    `xz` is never formed: the x rows of in_proj.weight are applied inside zigma_in_conv_x_proj_fwd (x lives in accumulators and
    LDS only), the z rows by a projection of their own.  Semantics of mamba_inner_tok(F.linear(h, in_proj_weight), ...):
    reference mamba_simple.py:290-294,362-395 + selective_scan_interface.py:296-365 (without out_proj)."""
    from .linear import linear, linear_eligible
    Di = in_proj_weight.shape[0] // 2
    w = conv1d_weight.reshape(Di, -1)
    u, x_dbl = in_conv_x_proj(h, in_proj_weight[:Di], w, conv1d_bias, x_proj_weight, perm)
    w_z = in_proj_weight[Di:]
    if linear_eligible(h, w_z, None, prefer_own=True):
        z = linear(h, w_z, silu_from_col=0 if Z_HALF_SILU_IN_EPILOGUE else None)
        zact = Z_HALF_SILU_IN_EPILOGUE
    else:
        z, zact = F.linear(h, w_z), False
    return _inner_tok_tail(u, x_dbl, z, delta_proj_weight, A, D, delta_bias, perm, out_rows, None, None, delta_softplus, out, 0, zact)
