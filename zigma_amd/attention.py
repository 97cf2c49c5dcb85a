"""Cross-attention core on the HIP kernel (zigma_cross_attn_fwd): softmax(scale * Q K^T) V over a short context.

Mirrors the F.scaled_dot_product_attention call inside the reference's CrossAttention.forward (model_zigma.py:113-127)
for the shapes ZigMa produces (8 heads x 64, 77 text tokens).  Forward only: when autograd is recording, or for operands
outside the kernel's limits (bf16, head_dim 64, n_ctx <= 128, 16-byte aligned rows), the caller keeps torch's SDPA.
"""
import os

import torch

from . import _lib

# to_q + attention core as ONE kernel (q never reaches memory).  Built, parity-tested — and no faster than the library to_q GEMM
# followed by cross_attn_kernel (84.6-92.5 us against 84.6-90.5, tools/q_attn_probe.py: the attention phase inside the GEMM
# epilogue runs at the 2 waves per SIMD of the 230-register projection kernel and takes as long as the stand-alone, HBM-bound
# attention kernel; what it saves in q traffic it loses there).  Off unless ZIGMA_Q_ATTN=1.
USE_Q_ATTN = os.environ.get("ZIGMA_Q_ATTN", "0") == "1"


def cross_attn_eligible(q, k, v, heads):
    if not (q.is_cuda and q.dtype == torch.bfloat16 and k.dtype == q.dtype and v.dtype == q.dtype):
        return False
    if q.dim() != 3 or k.dim() != 3 or v.shape != k.shape or q.shape[2] != heads * 64 or k.shape[2] != heads * 64:
        return False
    if k.shape[1] > 128 or k.shape[1] < 1 or q.shape[0] != k.shape[0]:
        return False
    ok = lambda t: t.stride(2) == 1 and t.stride(1) % 8 == 0 and t.stride(0) % 8 == 0 and t.data_ptr() % 16 == 0
    return ok(q) and ok(k) and ok(v)


def cross_attn(q, k, v, heads, scale=None):
    """q: (B, L, H*64); k, v: (B, n_ctx, H*64) (row-strided views are fine) -> (B, L, H*64)."""
    dev = _lib.require_device(q, k, v)
    if not cross_attn_eligible(q, k, v, heads):
        raise RuntimeError("cross_attn: needs bf16 (B, L, H*64) / (B, n_ctx <= 128, H*64) operands with 16-byte aligned rows")
    Bsz, L, C = q.shape
    out = torch.empty(Bsz, L, C, device=q.device, dtype=q.dtype)
    P = _lib.XAttnParams()
    P.batch, P.seqlen, P.n_ctx, P.heads, P.head_dim = Bsz, L, k.shape[1], heads, 64
    P.dtype, P.flags, P.scale = _lib.dtype_id(q), 0, float(64 ** -0.5 if scale is None else scale)
    for name, t in (("q", q), ("k", k), ("v", v), ("o", out)):
        setattr(P, name + "_batch_stride", t.stride(0))
        setattr(P, name + "_row_stride", t.stride(1))
    P.q, P.k, P.v, P.out = _lib.ptr(q), _lib.ptr(k), _lib.ptr(v), _lib.ptr(out)
    _lib.call("zigma_cross_attn_fwd", P, dev)
    return out


def transpose_v(v, keys=96):
    """(B, n_ctx, C) -> (B, C, keys) zero padded: the V^T layout zigma_q_attn_fwd reads (rows contiguous over the keys)."""
    Bsz, NC, C = v.shape
    vt = torch.zeros(Bsz, C, keys, device=v.device, dtype=v.dtype)
    vt[:, :, :NC] = v.transpose(1, 2)
    return vt


def q_attn_eligible(x, wq, k, heads):
    """limits of zigma_q_attn_fwd: bf16, head_dim 64, heads * 64 % 256 == 0, query_dim % 64 == 0, seqlen % 256 == 0, n_ctx <= 80,
    contiguous x, 16-byte aligned rows; at least 16 384 query tokens (below that the 256-token tiles do not fill the chip)"""
    if not (x.is_cuda and x.dtype == torch.bfloat16 and wq.dtype == x.dtype and k.dtype == x.dtype):
        return False
    if x.dim() != 3 or not x.is_contiguous() or k.dim() != 3 or k.shape[0] != x.shape[0]:
        return False
    Bsz, L, E = x.shape
    n = heads * 64
    return (wq.shape == (n, E) and n % 256 == 0 and E % 64 == 0 and L % 256 == 0 and Bsz * L >= 16384 and 1 <= k.shape[1] <= 80
            and k.shape[2] == n and wq.stride(1) == 1 and wq.stride(0) % 8 == 0 and k.stride(2) == 1 and k.stride(1) % 8 == 0
            and k.stride(0) % 8 == 0 and x.data_ptr() % 16 == 0 and wq.data_ptr() % 16 == 0 and k.data_ptr() % 16 == 0
            and Bsz * L * E * 2 < 2 ** 31 and n * wq.stride(0) * 2 < 2 ** 31)


def q_attn(x, wq, k, vt, heads, scale=None):
    """out = softmax(scale * (x @ wq^T)_h k_h^T) v_h per head, (B, L, heads*64): to_q and the attention core in one kernel.
    x: (B, L, E) contiguous; wq: (heads*64, E); k: (B, n_ctx, heads*64); vt: transpose_v(v) = (B, heads*64, keys)."""
    dev = _lib.require_device(x, wq, k, vt)
    Bsz, L, E = x.shape
    n = heads * 64
    if vt.shape[:2] != (Bsz, n) or vt.stride(2) != 1 or vt.dtype != x.dtype:
        raise RuntimeError("vt must be (B, heads*64, keys) with contiguous keys in the dtype of x")
    out = torch.empty(Bsz, L, n, device=x.device, dtype=x.dtype)
    P = _lib.QAttnParams()
    P.batch, P.seqlen, P.n_ctx, P.heads, P.head_dim, P.k_dim = Bsz, L, k.shape[1], heads, 64, E
    P.vt_keys, P.dtype, P.flags, P.scale = vt.shape[2], _lib.dtype_id(x), 0, float(64 ** -0.5 if scale is None else scale)
    P.x_row_stride, P.w_row_stride, P.o_row_stride = E, wq.stride(0), n
    P.k_batch_stride, P.k_row_stride, P.vt_batch_stride, P.vt_row_stride = k.stride(0), k.stride(1), vt.stride(0), vt.stride(1)
    P.x, P.w, P.k, P.vt, P.out = _lib.ptr(x), _lib.ptr(wq), _lib.ptr(k), _lib.ptr(vt), _lib.ptr(out)
    _lib.call("zigma_q_attn_fwd", P, dev)
    return out
