"""Weight gradients of the block's projections for the training path: dW = dY^T X with the TOKENS as the contraction dimension
(M = batch x seqlen = 65 536 at the headline shape against outputs of 72 x 1280 ... 2560 x 640 elements).

The reference leaves these to autograd's linear backward (one GEMM each); as ONE library GEMM they are 5 ... 100 output tiles for 256
CUs and run at 0.03 ... 0.5 PFLOP/s (tools/wgrad_split_probe.py: in_proj 450 us, to_q 259 us, x_proj 227 us).  Split along the tokens
into S slabs as a BATCHED GEMM (the slab index is the library's batch dimension) the same products fill the chip; the S partial
results (bf16, like a GEMM output) are added in fp32: in_proj 332 us, out_proj 303 -> 149, to_q 259 -> 76, to_out 271 -> 66,
x_proj 227 -> 72, dt_proj 205 -> 64 us — 17 ms of a 122 ms training step.  `LinearTrainFn` is F.linear with that backward; since round 4 its forward product and dX run on the hand-written projection kernels
(zigma_linear_fwd) where one serves the shape — `OWN_TRAIN_GEMMS`."""
import os

import torch
import torch.nn.functional as F

SPLIT_WGRAD = True        # False: one GEMM (A/B in tools/train_probe.py)
# forward projection and dX = dY W of the training path on the hand-written MFMA kernels (zigma_linear_fwd: weight-stationary / tiled) where they
# serve the shape; dW stays the slab-wise batched product below.  "0": the library for both (A/B in tools/train_probe.py)
OWN_TRAIN_GEMMS = True


_BMM_OUT_DTYPE = {}      # (device type, index, dtype) -> bool


def _bmm_takes_out_dtype(x):
    """does torch.bmm accept out_dtype=float32 for x's dtype ON x's device?  Probed lazily, once per (device, dtype), with a 1 x 1 x 1 product there (an
    import-time probe on the CPU says nothing about the GPU backend).  Only the two answers that mean "no" are caught — a missing keyword (TypeError) and a
    backend that declines the combination (RuntimeError naming it); anything else (e.g. out of memory) propagates."""
    key = (x.device.type, x.device.index, x.dtype)
    hit = _BMM_OUT_DTYPE.get(key)
    if hit is None:
        a = torch.zeros(1, 1, 1, device=x.device, dtype=x.dtype)
        try:
            torch.bmm(a, a, out_dtype=torch.float32)
            hit = True
        except TypeError:
            hit = False
        except RuntimeError as e:
            msg = str(e).lower()
            if not any(w in msg for w in ("out_dtype", "not implemented", "not supported", "unsupported", "expected")):
                raise
            hit = False
        _BMM_OUT_DTYPE[key] = hit
    return hit


def _own_linear(x, weight, bias=None, transposed=False):
    """x @ weight^T (+ bias) on zigma_linear_fwd if a kernel of it serves the call (autograd is off in here), else None.
    transposed: the product wanted is x @ weight (dX = dY W) — the transposed copy of the weight is only made once a kernel is known to
    take the call (eligibility is decided on shapes, dtypes and alignment, which the copy does not change)"""
    from . import routing
    from .linear import linear, linear_eligible, linear_ws_eligible
    if not OWN_TRAIN_GEMMS or routing.POLICY == "off" or not x.is_cuda or x.dtype != torch.bfloat16 or weight.dtype != torch.bfloat16:
        return None
    if transposed:
        wt = _ContiguousLike(weight)
        if not ((bias is None and linear_ws_eligible(x, wt)) or linear_eligible(x, wt, bias)):
            return None
        weight = weight.t().contiguous()
    if bias is None and linear_ws_eligible(x, weight):
        return linear(x, weight, weight_stationary=True)
    if linear_eligible(x, weight, bias):
        return linear(x, weight, bias)
    return None


class _ContiguousLike:
    """Stand-in for `weight.t().contiguous()` in the eligibility predicates of linear.py, which read device, dtype, shape, strides and the
    16-byte alignment of data_ptr (a fresh allocation is aligned) — so the decision can be taken before the copy is made."""

    def __init__(self, weight):
        self.is_cuda, self.dtype, self.device, self.requires_grad = weight.is_cuda, weight.dtype, weight.device, False
        self.shape = torch.Size((weight.shape[1], weight.shape[0]))

    def stride(self, i=None):
        st = (self.shape[1], 1)
        return st if i is None else st[i]

    def dim(self):
        return 2

    def data_ptr(self):
        return 0

    def is_contiguous(self):
        return True


def _slabs(m, n, k):
    """number of token slabs: enough (128 x 128)-tile workgroups for ~2 per CU, a power of two dividing m, slabs of >= 256 rows"""
    tiles = max(1, -(-n // 128) * -(-k // 128))
    s = 8
    while s < 64 and s * tiles < 512:
        s *= 2
    while s > 1 and (m % s or (m // s) < 256 or (m // s) % 8):
        s //= 2
    return s


def wgrad(dy2, x2):
    """dy2 (m, n), x2 (m, k), same 16-bit dtype -> dy2^T x2 (n, k) in that dtype (fp32 sum of the slab products)"""
    m, n = dy2.shape
    k = x2.shape[1]
    s = _slabs(m, n, k) if (SPLIT_WGRAD and dy2.is_cuda and dy2.dtype in (torch.bfloat16, torch.float16)) else 1
    if s <= 1:
        return dy2.t() @ x2
    dy3 = dy2.reshape(s, m // s, n)
    x3 = x2.reshape(s, m // s, k)
    # fp32 slab partials (one rounding at the end, like the single GEMM of the reference; a 16-bit partial could also overflow in fp16)
    if _bmm_takes_out_dtype(dy2):
        part = torch.bmm(dy3.transpose(1, 2), x3, out_dtype=torch.float32)
    else:                                  # (a torch without out_dtype on bmm)
        part = torch.bmm(dy3.transpose(1, 2), x3).float()
    return part.sum(0).to(dy2.dtype)


class LinearTrainFn(torch.autograd.Function):
    """F.linear whose weight gradient is formed slab-wise (see the module docstring); dX and dbias as autograd forms them."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        out = _own_linear(x, weight, bias)
        return F.linear(x, weight, bias) if out is None else out

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        dy2 = dy.reshape(-1, dy.shape[-1])
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dyc = dy2 if dy2.is_contiguous() else dy2.contiguous()
            dx = _own_linear(dyc, weight, transposed=True)         # dX = dY W = dY (W^T)^T: the same kernels on the transposed weight (a few MB)
            dx = (dy2 @ weight).view(x.shape) if dx is None else dx.view(x.shape)
        if ctx.needs_input_grad[1]:
            dw = wgrad(dy2, x.reshape(-1, x.shape[-1]))
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = dy2.sum(0)
        return dx, dw, db


def linear_train(x, weight, bias=None):
    """differentiable projection of the training path: the slab-wise weight gradient where it pays (many tokens), F.linear otherwise"""
    if (x.is_cuda and x.dtype in (torch.bfloat16, torch.float16) and weight.dtype == x.dtype and (bias is None or bias.dtype == x.dtype)
            and x.numel() // x.shape[-1] >= 4096 and torch.is_grad_enabled() and not torch.is_autocast_enabled()):
        return LinearTrainFn.apply(x, weight, bias)
    return F.linear(x, weight, bias)
