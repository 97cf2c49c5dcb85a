"""Fused residual-add + LayerNorm / RMSNorm op layer on the HIP kernel (no Triton).

Mirrors the reference's `dis_mamba/mamba_ssm/ops/triton/layernorm.py` forward API:
    layer_norm_fn / rms_norm_fn  (:464-478),  RMSNorm  (:481-503)
and adds `block_norm`, the ZigMa-block form with the gated branch add and adaLN modulate folded in.
"""
import torch

from . import _lib


def _norm_call(x2, weight, bias, residual2, eps, is_rms, residual_dtype, *, rows_per_batch=None, branch=None,
               gate=None, x_out=None, shift=None, scale=None, want_y=True, want_res=None):
    dev = _lib.require_device(x2, weight, bias, residual2, branch, gate, shift, scale)
    rows, cols = x2.shape
    P = _lib.NormParams()
    P.rows, P.cols = rows, cols
    P.rows_per_batch = rows_per_batch or max(rows, 1)
    P.is_rms, P.eps, P.flags = int(is_rms), float(eps), 0
    P.x_dtype = _lib.dtype_id(x2)
    P.x, P.x_row_stride = _lib.ptr(x2), x2.stride(0)
    y = y_mod = res_out = None
    if want_y:
        y = torch.empty(rows, cols, device=x2.device, dtype=x2.dtype)
        P.y_out, P.y_row_stride = _lib.ptr(y), y.stride(0)
    if residual2 is not None:
        P.residual, P.res_row_stride = _lib.ptr(residual2), residual2.stride(0)
        residual_dtype = residual2.dtype
    P.res_dtype = _lib._DT.get(residual_dtype if residual_dtype is not None else x2.dtype)
    if want_res is None:
        want_res = residual2 is not None or (residual_dtype is not None and residual_dtype != x2.dtype)
    if want_res:
        res_out = torch.empty(rows, cols, device=x2.device, dtype=residual_dtype or x2.dtype)
        P.residual_out, P.res_out_row_stride = _lib.ptr(res_out), res_out.stride(0)
    P.w_dtype = _lib.dtype_id(weight) if weight is not None else P.x_dtype
    if weight is not None:
        P.weight = _lib.ptr(weight)
    if bias is not None:
        if weight is not None and bias.dtype != weight.dtype:
            raise RuntimeError("bias must have the dtype of weight")
        P.bias = _lib.ptr(bias)
    P.mod_dtype = P.x_dtype
    if branch is not None:
        P.branch, P.branch_row_stride = _lib.ptr(branch), branch.stride(0)
        P.gate, P.mod_batch_stride = _lib.ptr(gate), gate.stride(0)
        if x_out is not None:
            P.x_out, P.x_out_row_stride = _lib.ptr(x_out), x_out.stride(0)
    if shift is not None:
        y_mod = torch.empty(rows, cols, device=x2.device, dtype=x2.dtype)
        P.shift, P.scale = _lib.ptr(shift), _lib.ptr(scale)
        if shift.stride(0) != scale.stride(0) or (gate is not None and gate.stride(0) != shift.stride(0)):
            raise RuntimeError("gate / shift / scale must share a row pitch")
        P.mod_batch_stride = shift.stride(0)
        P.y_mod, P.y_mod_row_stride = _lib.ptr(y_mod), y_mod.stride(0)
    _lib.call("zigma_add_norm_fwd", P, dev)
    return y, res_out, y_mod


def _flat(t):
    t2 = t.reshape(-1, t.shape[-1])
    return t2 if t2.stride(-1) == 1 else t2.contiguous()


def layer_norm_fn(x, weight, bias, residual=None, eps=1e-6, prenorm=False, residual_in_fp32=False,
                  is_rms_norm=False):
    """y = norm(x + residual) * weight + bias, statistics in float32.  Returns y, or (y, x + residual) when
    `prenorm`; the returned residual has residual.dtype, else float32 if `residual_in_fp32`, else x.dtype —
    the conventions of LayerNormFn.forward (layernorm.py:380-422)."""
    shape = x.shape
    x2 = _flat(x)
    r2 = None
    if residual is not None:
        if residual.shape != shape:
            raise RuntimeError("residual must have the shape of x")
        r2 = _flat(residual)
    res_dtype = residual.dtype if residual is not None else (torch.float32 if residual_in_fp32 else None)
    w = weight.contiguous() if weight is not None else None
    b = bias.contiguous() if bias is not None else None
    y, res_out, _ = _norm_call(x2, w, b, r2, eps, is_rms_norm, res_dtype)
    y = y.reshape(shape)
    if not prenorm:
        return y
    return y, (res_out.reshape(shape) if res_out is not None else x)


def rms_norm_fn(x, weight, bias, residual=None, prenorm=False, residual_in_fp32=False, eps=1e-6):
    return layer_norm_fn(x, weight, bias, residual, eps, prenorm, residual_in_fp32, True)


class RMSNorm(torch.nn.Module):
    def __init__(self, hidden_size, eps=1e-5, device=None, dtype=None):
        super().__init__()
        self.eps = eps
        self.weight = torch.nn.Parameter(torch.empty(hidden_size, device=device, dtype=dtype))
        self.register_parameter("bias", None)
        self.reset_parameters()

    def reset_parameters(self):
        torch.nn.init.ones_(self.weight)

    def forward(self, x, residual=None, prenorm=False, residual_in_fp32=False):
        # (the reference forwards an extra is_rms_norm= kwarg here that rms_norm_fn does not accept,
        #  layernorm.py:493-503; this is the intended behaviour)
        return rms_norm_fn(x, self.weight, self.bias, residual=residual, prenorm=prenorm,
                           residual_in_fp32=residual_in_fp32, eps=self.eps)


def block_norm(x, weight, bias, residual, eps, is_rms, *, residual_in_fp32=True, branch=None, gate=None,
               shift=None, scale=None, want_x=False, want_y=True, want_res_out=True):
    """One launch for the glue around a ZigMa sub-layer (model_zigma.py:415-458), on (B, L, E) tensors:

        xe  = x + gate[:, None] * branch          (if branch is given; the previous sub-layer's gated output)
        res = xe + residual                       (if residual is given)        -> returned (fp32)
        y   = norm(res) * weight (+ bias)                                        -> returned if want_y
        ym  = y * (1 + scale[:, None]) + shift[:, None]   (if shift is given)   -> returned
    Returns (xe or None, res or None, y or None, ym or None)."""
    Bsz, L, E = x.shape
    x2 = _flat(x)
    xe = torch.empty_like(x2) if (branch is not None and want_x) else None
    res_dtype = residual.dtype if residual is not None else (torch.float32 if residual_in_fp32 else None)
    # the residual stream comes back whenever the kernel changes it (add, cast or gated branch); otherwise the
    # reference hands x itself back (layernorm.py:177) and so do we
    want_res = want_res_out and (residual is not None or branch is not None or
                                 (res_dtype is not None and res_dtype != x2.dtype))
    y, res_out, ym = _norm_call(x2, weight, bias, _flat(residual) if residual is not None else None, eps, is_rms,
                                res_dtype, rows_per_batch=L, branch=_flat(branch) if branch is not None else None,
                                gate=gate, x_out=xe, shift=shift, scale=scale, want_y=want_y, want_res=want_res)
    if want_res_out and res_out is None:
        res_out = x2
    rs = lambda t: None if t is None else t.reshape(Bsz, L, E)
    return rs(xe), rs(res_out), rs(y), rs(ym)
