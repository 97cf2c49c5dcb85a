"""Cross-attention core on the HIP kernel (zigma_cross_attn_fwd): softmax(scale * Q K^T) V over a short context.

Mirrors the F.scaled_dot_product_attention call inside the reference's CrossAttention.forward (model_zigma.py:113-127)
for the shapes ZigMa produces (8 heads x 64, 77 text tokens).  `cross_attn` is the forward kernel; `cross_attn_train` is its
autograd form for the training path: forward and backward are both hand-written HIP kernels (zigma_cross_attn_bwd recomputes the
probabilities: with 77 keys they are 80 MB per layer, recomputing them on the matrix cores is cheaper than storing them and than any
flash machinery) — torch's fused SDPA is an AOT-Triton kernel on ROCm, which the north star rules out.
Operands outside the kernels' limits (bf16, head_dim 64, n_ctx <= 128, 16-byte aligned rows) take the same math in plain torch ops.
"""
import torch

from . import _lib

def cross_attn_eligible(q, k, v, heads):
    if not (q.is_cuda and q.dtype == torch.bfloat16 and k.dtype == q.dtype and v.dtype == q.dtype):
        return False
    if q.dim() != 3 or k.dim() != 3 or v.shape != k.shape or q.shape[2] != heads * 64 or k.shape[2] != heads * 64:
        return False
    if k.shape[1] > 128 or k.shape[1] < 1 or q.shape[0] != k.shape[0] or q.shape[0] > 65535 or heads > 65535:      # (grid dims of the kernels)
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


def _heads(t, H):
    """(B, n, H * d) -> (B, H, n, d) view"""
    return t.view(t.shape[0], t.shape[1], H, -1).transpose(1, 2)


def _attention_math_plain(q, k, v, heads, scale):
    qh, kh, vh = _heads(q, heads), _heads(k, heads), _heads(v, heads)
    p = torch.softmax(torch.matmul(qh, kh.transpose(-1, -2)).float() * scale, dim=-1).to(q.dtype)
    return torch.matmul(p, vh).transpose(1, 2).reshape(q.shape)


ATTN_MATH_CHUNK_ELEMS = 2 ** 26      # fp32 probabilities alive at a time (256 MB): (B, H, L, n_ctx) is walked in batch chunks of this size


def _batch_chunks(q, k, heads):
    per_sample = max(heads * q.shape[1] * k.shape[1], 1)
    step = max(1, ATTN_MATH_CHUNK_ELEMS // per_sample)
    return [slice(i, min(i + step, q.shape[0])) for i in range(0, q.shape[0], step)]


class _AttentionMathFn(torch.autograd.Function):
    """attention_math under autograd without keeping the (B, H, L, n_ctx) probabilities of every layer alive: only q, k, v are
    saved, the backward recomputes the probabilities chunk by chunk (the memory behaviour of the reference's fused attention)."""

    @staticmethod
    def forward(ctx, q, k, v, heads, scale):
        ctx.save_for_backward(q, k, v)
        ctx.heads, ctx.scale = heads, scale
        out = torch.empty_like(q)
        for sl in _batch_chunks(q, k, heads):
            out[sl] = _attention_math_plain(q[sl], k[sl], v[sl], heads, scale)
        return out

    @staticmethod
    def backward(ctx, dout):
        q, k, v = ctx.saved_tensors
        dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
        for sl in _batch_chunks(q, k, ctx.heads):
            with torch.enable_grad():
                qc, kc, vc = (t[sl].detach().requires_grad_(True) for t in (q, k, v))
                o = _attention_math_plain(qc, kc, vc, ctx.heads, ctx.scale)
            dq[sl], dk[sl], dv[sl] = torch.autograd.grad(o, (qc, kc, vc), dout[sl])
        return dq, dk, dv, None, None


def attention_math(q, k, v, heads, scale):
    """softmax(scale q k^T) v per head with batched GEMMs and ATen ops (no fused SDPA): the fallback for operands the kernel does not
    take.  Walks the batch in chunks so that at most ATTN_MATH_CHUNK_ELEMS fp32 probabilities exist at a time, and under autograd
    saves q, k, v only (the probabilities are recomputed in the backward)."""
    if torch.is_grad_enabled() and (q.requires_grad or k.requires_grad or v.requires_grad):
        return _AttentionMathFn.apply(q, k, v, heads, scale)
    chunks = _batch_chunks(q, k, heads)
    if len(chunks) == 1:
        return _attention_math_plain(q, k, v, heads, scale)
    out = torch.empty_like(q)
    for sl in chunks:
        out[sl] = _attention_math_plain(q[sl], k[sl], v[sl], heads, scale)
    return out


def cross_attn_bwd(q, k, v, dout, heads, scale=None):
    """gradients of cross_attn(q, k, v) w.r.t. q, k, v given dout (zigma_cross_attn_bwd: the probabilities are recomputed, nothing of
    the forward is needed).  Returns (dq, dk, dv) in the operand dtype; dk / dv are accumulated in fp32."""
    dev = _lib.require_device(q, k, v, dout)
    if not (cross_attn_eligible(q, k, v, heads) and cross_attn_eligible(dout, k, v, heads) and dout.shape == q.shape):
        raise RuntimeError("cross_attn_bwd: needs bf16 (B, L, H*64) / (B, n_ctx <= 128, H*64) operands with 16-byte aligned rows")
    Bsz, L, C = q.shape
    NC = k.shape[1]
    dq = torch.empty(Bsz, L, C, device=q.device, dtype=q.dtype)
    chunks = _lib.lib().zigma_cross_attn_bwd_chunks(L)
    part = torch.empty(2, max(chunks, 1), Bsz, NC, C, device=q.device, dtype=torch.float32)
    P = _lib.XAttnBwdParams()
    P.batch, P.seqlen, P.n_ctx, P.heads, P.head_dim = Bsz, L, NC, heads, 64
    P.dtype, P.flags, P.scale, P.chunks = _lib.dtype_id(q), 0, float(64 ** -0.5 if scale is None else scale), chunks
    for name, t in (("q", q), ("k", k), ("v", v), ("do", dout), ("dq", dq)):
        setattr(P, name + "_batch_stride", t.stride(0))
        setattr(P, name + "_row_stride", t.stride(1))
    P.q, P.k, P.v, P.dout, P.dq = _lib.ptr(q), _lib.ptr(k), _lib.ptr(v), _lib.ptr(dout), _lib.ptr(dq)
    P.dk_part, P.dv_part = _lib.ptr(part[0]), _lib.ptr(part[1])
    if L == 0 or Bsz == 0:
        part.zero_()
    _lib.call("zigma_cross_attn_bwd", P, dev)
    dkv = (part[:, 0] if chunks <= 1 else part.sum(1)).to(q.dtype)
    return dq, dkv[0], dkv[1]


def cross_attn_bwd_math(q, k, v, dout, heads, scale):
    """the same gradients with batched GEMMs and ATen ops (probabilities recomputed in fp32, rounded to bf16 where the kernels round
    them): the checker of cross_attn_bwd in the tests and the A/B leg of tools/train_probe.py"""
    H = heads
    qh, kh, vh, doh = _heads(q, H), _heads(k, H), _heads(v, H), _heads(dout.contiguous(), H)
    p32 = torch.softmax(torch.matmul(qh, kh.transpose(-1, -2)).float() * scale, dim=-1)          # (B, H, L, n_ctx)
    p = p32.to(q.dtype)
    dv = torch.matmul(p.transpose(-1, -2), doh)                                                    # (B, H, n_ctx, d)
    dp = torch.matmul(doh, vh.transpose(-1, -2)).float()
    ds = (p32 * (dp - (dp * p32).sum(-1, keepdim=True)) * scale).to(q.dtype)
    dq = torch.matmul(ds, kh)                                                                      # (B, H, L, d)
    dk = torch.matmul(ds.transpose(-1, -2), qh)
    back = lambda t, like: t.transpose(1, 2).reshape(like.shape)
    return back(dq, q), back(dk, k), back(dv, v)


BWD_KERNEL = True      # False: cross_attn_bwd_math (A/B in tools/train_probe.py)


class CrossAttnFn(torch.autograd.Function):
    """cross_attn with a backward (reference: autograd through F.scaled_dot_product_attention, model_zigma.py:123):
        P = softmax(scale Q K^T);  dV = P^T dO;  dP = dO V^T;  dS = P * (dP - rowsum(dP * P)) * scale;  dQ = dS K;  dK = dS^T Q
    Both directions are hand-written kernels (zigma_cross_attn_fwd / zigma_cross_attn_bwd); only q, k, v are saved."""

    @staticmethod
    def forward(ctx, q, k, v, heads, scale):
        ctx.save_for_backward(q, k, v)
        ctx.heads, ctx.scale = heads, scale
        return cross_attn(q, k, v, heads, scale)

    @staticmethod
    def backward(ctx, do):
        q, k, v = ctx.saved_tensors
        do = do if do.stride(-1) == 1 and do.stride(1) % 8 == 0 and do.stride(0) % 8 == 0 and do.data_ptr() % 16 == 0 else do.contiguous()
        fn = cross_attn_bwd if BWD_KERNEL else cross_attn_bwd_math
        return (*fn(q, k, v, do, ctx.heads, ctx.scale), None, None)


def cross_attn_train(q, k, v, heads, scale=None):
    """differentiable cross_attn (the training path)"""
    return CrossAttnFn.apply(q, k, v, heads, float(64 ** -0.5 if scale is None else scale))
