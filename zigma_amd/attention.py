"""Cross-attention core on the HIP kernel (zigma_cross_attn_fwd): softmax(scale * Q K^T) V over a short context.

Mirrors the F.scaled_dot_product_attention call inside the reference's CrossAttention.forward (model_zigma.py:113-127)
for the shapes ZigMa produces (8 heads x 64, 77 text tokens).  Forward only: when autograd is recording, or for operands
outside the kernel's limits (bf16, head_dim 64, n_ctx <= 128, 16-byte aligned rows), the caller keeps torch's SDPA.
"""
import torch

from . import _lib

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
