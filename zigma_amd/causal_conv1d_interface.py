"""Depthwise causal conv1d op layer on the HIP kernel.

Mirrors the reference's `dis_causal_conv1d/causal_conv1d/causal_conv1d_interface.py`:
    causal_conv1d_fwd  <-> causal_conv1d_cuda.causal_conv1d_fwd   (causal_conv1d.cpp:130-189)
    causal_conv1d_fn   <-> causal_conv1d_fn                       (causal_conv1d_interface.py:37-46)
Forward only (bwd / update are later scope rows).
"""
import torch

from . import _lib


def causal_conv1d_raw(x, weight, bias, silu, *, out=None, x_row_index=None, reset_period=0):
    """x, out: logical (batch, dim, seqlen) views, any strides.  weight (dim, width), bias (dim,) or None."""
    dev = _lib.require_device(x, weight, bias, out, x_row_index)
    if x.dim() != 3:
        raise RuntimeError("x must be (batch, dim, seqlen)")
    batch, dim, L = x.shape
    if weight.dim() != 2 or weight.shape[0] != dim:
        raise RuntimeError("weight must be (dim, width)")
    width = weight.shape[1]
    if not 2 <= width <= 4:
        raise RuntimeError("causal_conv1d only supports width between 2 and 4")
    if bias is not None:
        if bias.dtype != weight.dtype or bias.shape != (dim,):
            raise RuntimeError("bias must be (dim,) with the dtype of weight")
        if bias.stride(0) != 1:
            bias = bias.contiguous()
    if out is None:
        out = torch.empty_like(x)
    elif out.shape != x.shape or out.dtype != x.dtype:
        raise RuntimeError("out must match x")
    P = _lib.ConvParams()
    P.batch, P.dim, P.seqlen, P.width = batch, dim, L, width
    P.silu_activation = int(bool(silu))
    P.io_dtype, P.w_dtype, P.flags = _lib.dtype_id(x), _lib.dtype_id(weight), 0
    P.x_batch_stride, P.x_c_stride, P.x_l_stride = x.stride()
    P.weight_c_stride, P.weight_width_stride = weight.stride()
    P.out_batch_stride, P.out_c_stride, P.out_l_stride = out.stride()
    P.x, P.weight, P.bias, P.out = _lib.ptr(x), _lib.ptr(weight), _lib.ptr(bias), _lib.ptr(out)
    if x_row_index is not None:
        if x_row_index.dtype != torch.int32 or x_row_index.shape != (L,) or not x_row_index.is_contiguous():
            raise RuntimeError("x_row_index must be a contiguous int32 tensor of length seqlen")
        P.x_row_index = _lib.ptr(x_row_index)
    P.reset_period = int(reset_period)       # > 0: independent sequences of that many positions along seqlen
    _lib.call("zigma_causal_conv1d_fwd", P, dev)
    return out


def causal_conv1d_fwd(x, weight, bias_, silu_activation):
    """Drop-in for the extension entry `causal_conv1d_cuda.causal_conv1d_fwd(x, weight, bias, silu) -> out`."""
    return causal_conv1d_raw(x, weight, bias_, silu_activation)


def causal_conv1d_fn(x, weight, bias=None, activation=None):
    """x: (batch, dim, seqlen); weight: (dim, width); bias: (dim,); activation: None | "silu" | "swish"."""
    if activation not in [None, "silu", "swish"]:
        raise NotImplementedError("activation must be None, silu, or swish")
    return causal_conv1d_raw(x, weight, bias, activation in ["silu", "swish"])


def conv_bwd_tok(x, weight, bias, dout, silu, x_row_index=None, dx=None, reset_period=0):
    """Backward of the token-major causal conv (zigma_causal_conv1d_bwd; reference causal_conv1d_cuda.causal_conv1d_bwd,
    causal_conv1d.cpp:191-283).  x, dout: (batch, seqlen, dim), channel stride 1; dout in SCAN order; x is read through
    x_row_index as in the forward and dx is scattered back through it.  Returns dx (dtype of x), dweight (dim, width) f32,
    dbias (dim) f32 or None.  reset_period > 0: independent sequences of that many positions along seqlen, as in the forward."""
    dev = _lib.require_device(x, weight, bias, dout, x_row_index)
    Bsz, L, Dm = x.shape
    if dout.shape != x.shape or x.stride(2) != 1 or dout.stride(2) != 1 or dout.dtype != x.dtype:
        raise RuntimeError("x, dout must be (batch, seqlen, dim) with channel stride 1 and one dtype")
    w = weight.reshape(Dm, -1)
    if dx is None:
        dx = torch.empty(Bsz, L, Dm, device=x.device, dtype=x.dtype)
    elif dx.shape != x.shape or dx.stride(2) != 1 or dx.dtype != x.dtype:
        raise RuntimeError("dx must be (batch, seqlen, dim) with channel stride 1 and the dtype of x")
    if Bsz > 65535:          # one launch rides the batch in a 16-bit grid dimension: slices, parameter gradients summed
        dw = db = None
        for a in range(0, Bsz, 65535):
            b = min(a + 65535, Bsz)
            _, dwi, dbi = conv_bwd_tok(x[a:b], weight, bias, dout[a:b], silu, x_row_index, dx=dx[a:b], reset_period=reset_period)
            dw = dwi if dw is None else dw + dwi
            db = dbi if db is None or dbi is None else db + dbi
        return dx, dw, db
    dw = torch.zeros(Dm, w.shape[1], device=x.device, dtype=torch.float32)
    db = torch.zeros(Dm, device=x.device, dtype=torch.float32) if bias is not None else None
    P = _lib.ConvBwdParams()
    P.batch, P.dim, P.seqlen, P.width = Bsz, Dm, L, w.shape[1]
    P.silu_activation, P.io_dtype, P.w_dtype, P.flags = int(bool(silu)), _lib.dtype_id(x), _lib.dtype_id(w), 0
    P.x_batch_stride, P.x_l_stride = x.stride(0), x.stride(1)
    P.dout_batch_stride, P.dout_l_stride = dout.stride(0), dout.stride(1)
    P.dx_batch_stride, P.dx_l_stride = dx.stride(0), dx.stride(1)
    P.weight_c_stride, P.weight_width_stride = w.stride(0), w.stride(1)
    P.x, P.weight, P.bias, P.dout, P.dx = _lib.ptr(x), _lib.ptr(w), _lib.ptr(bias), _lib.ptr(dout), _lib.ptr(dx)
    P.dweight, P.dbias = _lib.ptr(dw), _lib.ptr(db)
    P.reset_period = int(reset_period)
    if bias is not None and bias.dtype != w.dtype:
        raise RuntimeError("bias must have the dtype of weight")
    if x_row_index is not None:
        if x_row_index.dtype != torch.int32 or x_row_index.shape != (L,) or not x_row_index.is_contiguous():
            raise RuntimeError("x_row_index must be a contiguous int32 (seqlen,) table")
        P.x_row_index = _lib.ptr(x_row_index)
    ws = _lib.workspace("zigma_causal_conv1d_bwd", P, dev)
    _lib.call("zigma_causal_conv1d_bwd", P, dev)
    del ws
    return dx, dw, db
