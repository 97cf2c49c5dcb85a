"""Fused residual-add + LayerNorm / RMSNorm op layer on the HIP kernel (no Triton).

Mirrors the reference's `dis_mamba/mamba_ssm/ops/triton/layernorm.py` forward API:
    layer_norm_fn / rms_norm_fn  (:464-478),  RMSNorm  (:481-503)
and adds `block_norm`, the ZigMa-block form with the gated branch add and adaLN modulate folded in.
"""
import torch


from . import _lib

NORM_FLAGS = 0      # 1: one row per wave even where four fit (A/B probe; tools set it directly)


def _norm_call(x2, weight, bias, residual2, eps, is_rms, residual_dtype, *, rows_per_batch=None, branch=None,
               gate=None, x_out=None, shift=None, scale=None, want_y=True, want_res=None):
    dev = _lib.require_device(x2, weight, bias, residual2, branch, gate, shift, scale)
    rows, cols = x2.shape
    P = _lib.NormParams()
    P.rows, P.cols = rows, cols
    P.rows_per_batch = rows_per_batch or max(rows, 1)
    P.is_rms, P.eps, P.flags = int(is_rms), float(eps), NORM_FLAGS
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


def _layer_norm_fwd(x, weight, bias, residual, eps, prenorm, residual_in_fp32, is_rms_norm):
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
    return y.reshape(shape), (res_out.reshape(shape) if res_out is not None else x)


def norm_bwd(xsum, weight, dy, dresidual_out, eps, is_rms, *, x_dtype, want_dx=True, want_dres=False, has_bias=False):
    """zigma_add_norm_bwd (reference _layer_norm_bwd, layernorm.py:275-377).  xsum: the normalised tensor (..., cols);
    returns dx (x_dtype) | None, dresidual (xsum.dtype) | None, dweight f32 | None, dbias f32 | None."""
    dev = _lib.require_device(xsum, weight, dy, dresidual_out)
    shape = xsum.shape
    s2, dy2 = _flat(xsum), _flat(dy)
    rows, cols = s2.shape
    P = _lib.NormBwdParams()
    P.rows, P.cols, P.is_rms, P.eps, P.flags = rows, cols, int(is_rms), float(eps), 0
    P.x_dtype, P.res_dtype = _lib._DT[x_dtype], _lib.dtype_id(s2)
    P.w_dtype = _lib.dtype_id(weight) if weight is not None else P.x_dtype
    P.xsum, P.xsum_row_stride, P.dy, P.dy_row_stride = _lib.ptr(s2), s2.stride(0), _lib.ptr(dy2), dy2.stride(0)
    if dy2.dtype != x_dtype:
        raise RuntimeError("dy must have the dtype of the normalised output")
    dx = dres = dw = db = None
    if weight is not None:
        weight = weight.contiguous()          # kept alive in this local until after the launch
        P.weight = _lib.ptr(weight)
        dw = torch.zeros(cols, device=xsum.device, dtype=torch.float32)
        P.dweight = _lib.ptr(dw)
    if has_bias:
        db = torch.zeros(cols, device=xsum.device, dtype=torch.float32)
        P.dbias = _lib.ptr(db)
    if dresidual_out is not None:
        dr2 = _flat(dresidual_out)
        if dr2.dtype != s2.dtype:
            raise RuntimeError("dresidual_out must have the dtype of the residual stream")
        P.dresidual_out, P.dres_out_row_stride = _lib.ptr(dr2), dr2.stride(0)
    if want_dx:
        dx = torch.empty(rows, cols, device=xsum.device, dtype=x_dtype)
        P.dx, P.dx_row_stride = _lib.ptr(dx), dx.stride(0)
    if want_dres:
        dres = torch.empty(rows, cols, device=xsum.device, dtype=s2.dtype)
        P.dresidual, P.dres_row_stride = _lib.ptr(dres), dres.stride(0)
    ws = _lib.workspace("zigma_add_norm_bwd", P, dev)
    _lib.call("zigma_add_norm_bwd", P, dev)
    del ws
    return (None if dx is None else dx.reshape(shape)), (None if dres is None else dres.reshape(shape)), dw, db


class LayerNormFn(torch.autograd.Function):
    """Autograd wrapper of the HIP forward / backward kernels; same contract as the reference's LayerNormFn
    (layernorm.py:380-461): saves the normalised tensor (x, or x + residual), recomputes the statistics."""

    @staticmethod
    def forward(ctx, x, weight, bias, residual, eps, prenorm, residual_in_fp32, is_rms_norm):
        y, res_out = _layer_norm_fwd(x, weight, bias, residual, eps, prenorm, residual_in_fp32, is_rms_norm)
        ctx.save_for_backward(res_out, weight)
        ctx.eps, ctx.is_rms, ctx.prenorm = eps, is_rms_norm, prenorm
        ctx.x_dtype, ctx.has_residual, ctx.has_bias = x.dtype, residual is not None, bias is not None
        ctx.w_dtype = weight.dtype if weight is not None else None
        return (y, res_out) if prenorm else y

    @staticmethod
    def backward(ctx, dy, *rest):
        xsum, weight = ctx.saved_tensors
        dres_out = rest[0] if ctx.prenorm and rest and rest[0] is not None else None
        if dres_out is not None:
            dres_out = dres_out.contiguous()
        dx, dres, dw, db = norm_bwd(xsum, weight, dy.contiguous(), dres_out, ctx.eps, ctx.is_rms, x_dtype=ctx.x_dtype,
                                    want_dx=True, want_dres=ctx.has_residual, has_bias=ctx.has_bias)
        dw = None if dw is None else dw.to(ctx.w_dtype)
        db = None if db is None else db.to(ctx.w_dtype)
        return dx, dw, db, dres, None, None, None, None


def layer_norm_fn(x, weight, bias, residual=None, eps=1e-6, prenorm=False, residual_in_fp32=False,
                  is_rms_norm=False):
    """y = norm(x + residual) * weight + bias, statistics in float32.  Returns y, or (y, x + residual) when
    `prenorm`; the returned residual has residual.dtype, else float32 if `residual_in_fp32`, else x.dtype —
    the conventions of LayerNormFn.forward (layernorm.py:380-422).  Differentiable (HIP backward) when autograd is
    recording and any operand requires grad."""
    if torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in (x, weight, bias, residual)):
        return LayerNormFn.apply(x, weight, bias, residual, eps, prenorm, residual_in_fp32, is_rms_norm)
    y, res_out = _layer_norm_fwd(x, weight, bias, residual, eps, prenorm, residual_in_fp32, is_rms_norm)
    return (y, res_out) if prenorm else y


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


def glue_bwd_eligible(dy, a, s):
    """limits of zigma_scale_reduce_bwd: bf16 (B, L, E) rows in one pitch, L % 64 == 0, E % 128 == 0, E <= 8192, 16-byte aligned"""
    if not (dy.is_cuda and dy.dtype == torch.bfloat16 and a.dtype == dy.dtype and s.dtype == dy.dtype and dy.dim() == 3 and a.shape == dy.shape):
        return False
    Bsz, L, E = dy.shape
    ok = lambda t: t.stride(2) == 1 and t.stride(1) % 8 == 0 and t.stride(0) == L * t.stride(1) and t.data_ptr() % 16 == 0
    return (L % 64 == 0 and E % 128 == 0 and E <= 8192 and ok(dy) and ok(a) and s.shape == (Bsz, E) and s.stride(1) == 1 and s.stride(0) % 8 == 0
            and s.data_ptr() % 16 == 0)


def scale_reduce_bwd(dy, a, s, s_add=0.0, want_out=True, want_sum=False):
    """One pass for the backward of the block's elementwise glue (zigma_scale_reduce_bwd):
        out = dy * (s[:, None] + s_add);  r1 = sum_L dy * a;  r2 = sum_L dy (if want_sum)      ->  (out | None, r1, r2 | None), r in dy.dtype"""
    dev = _lib.require_device(dy, a, s)
    Bsz, L, E = dy.shape
    nch = L // 64
    out = torch.empty(Bsz, L, E, device=dy.device, dtype=dy.dtype) if want_out else None
    r1 = torch.empty(Bsz, nch, E, device=dy.device, dtype=torch.float32)
    r2 = torch.empty(Bsz, nch, E, device=dy.device, dtype=torch.float32) if want_sum else None
    P = _lib.GlueBwdParams()
    P.rows, P.cols, P.rows_per_batch, P.dtype, P.flags, P.s_add = Bsz * L, E, L, _lib.dtype_id(dy), 0, float(s_add)
    P.dy_row_stride, P.a_row_stride, P.out_row_stride, P.s_batch_stride = dy.stride(1), a.stride(1), E, s.stride(0)
    P.dy, P.a, P.s, P.out, P.r1, P.r2 = _lib.ptr(dy), _lib.ptr(a), _lib.ptr(s), _lib.ptr(out), _lib.ptr(r1), _lib.ptr(r2)
    _lib.call("zigma_scale_reduce_bwd", P, dev)
    return out, r1.sum(1).to(dy.dtype), None if r2 is None else r2.sum(1).to(dy.dtype)
