"""Dense projections on the hand-written MFMA kernel (zigma_linear_fwd): in_proj / out_proj of the Mamba mixer and
to_q / to_out of the cross-attention (reference call sites mamba_simple.py:290-294, selective_scan_interface.py:365,
model_zigma.py:104-135, all `F.linear`).  Under autograd the callers go through wgrad.LinearTrainFn, whose forward product and dX use the same kernels (its own eligibility
check runs with autograd off); linear_eligible itself refuses tensors that require grad."""
import torch

from . import _lib
from . import routing


def linear_eligible(x, weight, bias=None):
    """LIMITS of zigma_linear_fwd's tiled kernels on these tensors (no policy — which projection runs where is zigma_amd/routing.py): bf16 on the
    device, k % 64 == 0, n % 128 == 0, tokens % 8 == 0, aligned contiguous rows, no autograd"""
    if not (x.is_cuda and x.dtype == torch.bfloat16 and weight.dtype == torch.bfloat16):
        return False
    if torch.is_grad_enabled() and (x.requires_grad or weight.requires_grad or (bias is not None and bias.requires_grad)):
        return False
    n, k = weight.shape
    if k % 64 or n % 128 or x.shape[-1] != k or x.stride(-1) != 1 or weight.stride(1) != 1:
        return False
    if bias is not None and (bias.dtype != torch.bfloat16 or bias.stride(0) != 1 or n > 4096 or bias.data_ptr() % 4):
        return False                                     # (the kernel stages the bias vector in 8 KB of LDS)
    m = x.numel() // k
    if m % 8 or m == 0:
        return False
    if x.dim() > 2 and not x.is_contiguous():
        return False
    if x.stride(-2) % 8 or weight.stride(0) % 8 or x.data_ptr() % 16 or weight.data_ptr() % 16:
        return False
    return m * x.stride(-2) * 2 < 2 ** 31 and n * weight.stride(0) * 2 < 2 ** 31 and 256 * n * 2 < 2 ** 31


LINEAR_WS_FLAG = 0x4000          # zigma_linear_params_t.flags: ZIGMA_LINEAR_WS (csrc/linear_ws.hip)


def linear_ws_eligible(x, weight, bias=None):
    """limits of the weight-stationary kernel (csrc/linear_ws.hip: a W panel lives in the registers of a workgroup, only the tokens stream): bf16, no
    bias; k = 512 or 640 with 256-feature panels (n % 256 == 0), or k = 1280 / 1536 with 128-feature panels (n % 128 == 0: the out_proj shapes, used
    below the tiled 4-wave kernel's floor); n <= 8192, tokens % 512 == 0 and enough of them for every workgroup of an XCD to own a tile, x rows a multiple
    of 128 elements apart — on top of linear_eligible's alignment rules."""
    if bias is not None or not linear_eligible(x, weight, None):
        return False
    n, k = weight.shape
    # (routing.serves_ws mirrors linear_ws_panel of the C side, including its 32-panel limit: n <= 4096 for the 128-feature form — ADVICE r5)
    return routing.serves_ws(x.numel() // k, n, k) and x.stride(-2) % 128 == 0


LINEAR_SM_FLAG = 0x8000          # zigma_linear_params_t.flags: ZIGMA_LINEAR_SM (csrc/linear_sm.hip)


def linear_sm_eligible(x, weight, bias=None):
    """limits of the few-token tiled kernel (csrc/linear_sm.hip: tiles of 128 tokens x n / 4 features, one per workgroup — 8192 tokens x 640
    features are exactly 256 tiles): bf16, k % 64 == 0 and k >= 128, tokens % 128 == 0 (tiles of 160, 192 or 128 features), an optional bf16
    bias on an 8-byte boundary — on top of linear_eligible's alignment rules.  The gated residual epilogue: gated_residual_eligible, as for the
    tiled kernels."""
    if not linear_eligible(x, weight, bias):
        return False
    if bias is not None and bias.data_ptr() % 8:
        return False
    n, k = weight.shape
    m = x.numel() // k
    return k >= 128 and m % 128 == 0 and m >= 128        # (n % 128 == 0 by linear_eligible)


def linear(x, weight, bias=None, silu_from_col=None, out=None, _probe_flags=0, residual=None, gate=None, weight_stationary=False, few_tokens=False):
    """out = x @ weight.T (+ bias); output columns >= silu_from_col (a multiple of 32) leave as silu(.).
    residual (same shape as the result) + gate (batch, n): out = residual + gate[b] * bf16(x @ weight.T + bias) in the kernel's
    epilogue (the gated branch add of the reference's Block, model_zigma.py:447-449); x must then be (batch, rows, k) with
    rows % 256 == 0.  weight_stationary: the csrc/linear_ws.hip kernel (linear_ws_eligible shapes only; fails otherwise); few_tokens: the
    csrc/linear_sm.hip kernel (linear_sm_eligible shapes only)."""
    dev = _lib.require_device(x, weight, bias, out, residual, gate)
    lead, k = x.shape[:-1], x.shape[-1]
    x2 = x.reshape(-1, k)
    n = weight.shape[0]
    if out is None:
        out = torch.empty(x2.shape[0], n, device=x.device, dtype=x.dtype)
    o2 = out if out.dim() == 2 else out.view(-1, n)            # a view: the kernel writes through the row pitch
    P = _lib.LinearParams()
    P.m, P.n, P.k, P.dtype, P.flags = x2.shape[0], n, k, _lib.dtype_id(x), int(_probe_flags) | (LINEAR_WS_FLAG if weight_stationary else LINEAR_SM_FLAG if few_tokens else 0)
    P.silu_from_col = n if silu_from_col is None else int(silu_from_col)
    P.x_row_stride, P.w_row_stride, P.out_row_stride = x2.stride(0), weight.stride(0), o2.stride(0)
    P.x, P.w, P.bias, P.out = _lib.ptr(x2), _lib.ptr(weight), _lib.ptr(bias), _lib.ptr(o2)
    if residual is not None:
        if gate is None or x.dim() != 3 or residual.shape != (*lead, n) or residual.dtype != x.dtype or gate.dtype != x.dtype \
                or gate.shape != (x.shape[0], n) or gate.stride(1) != 1 or residual.stride(-1) != 1 or not residual_rows_ok(residual):
            raise RuntimeError("linear: residual (B, rows, n) with uniform row pitch and gate (B, n) rows in the dtype of x")
        P.residual, P.gate = _lib.ptr(residual), _lib.ptr(gate)
        P.res_row_stride, P.gate_batch_stride, P.rows_per_batch = residual.stride(1), gate.stride(0), x.shape[1]
    _lib.call("zigma_linear_fwd", P, dev)
    return out if out.dim() == len(lead) + 1 and out.shape[:-1] == lead else out.view(*lead, n)


def residual_rows_ok(residual):
    """(B, rows, n) whose rows of all samples form ONE sequence of rows of the same pitch"""
    return residual.dim() == 3 and residual.stride(0) == residual.shape[1] * residual.stride(1)


def gated_residual_eligible(x, residual, gate):
    """limits of the gated-residual epilogue: (B, rows % 256 == 0, k) input, bf16 residual rows 16-byte aligned in one pitch,
    gate rows 16-byte aligned"""
    return (x.dim() == 3 and x.shape[1] % 256 == 0 and residual.dtype == x.dtype and gate.dtype == x.dtype and residual_rows_ok(residual)
            and residual.stride(-1) == 1 and residual.stride(1) % 8 == 0 and residual.data_ptr() % 16 == 0
            and gate.dim() == 2 and gate.stride(1) == 1 and gate.stride(0) % 8 == 0 and gate.data_ptr() % 16 == 0
            and 256 * residual.stride(1) * 2 < 2 ** 31)


def project(role, x, weight, bias=None, residual=None, gate=None):
    """The block loop's projections through ONE dispatch (zigma_amd/routing.py): role in_proj | out_proj | to_q | to_out.
    residual (B, L, n) + gate (B, n): the result is residual + gate[:, None] * (x @ weight.T + bias) — in the serving kernel's epilogue where the table says
    so and the kernel's limits are met, as an addcmul behind the product otherwise.  Calls the own kernels cannot take (fp32 / fp16 models, CPU tensors,
    autograd) go to wgrad.linear_train (F.linear; under autograd with the slab-wise weight gradient)."""
    from .wgrad import linear_train
    n, k = weight.shape
    tokens = x.numel() // max(k, 1)
    own = linear_eligible(x, weight, bias)
    r = routing.route(role, tokens, n, k) if own else routing.Route("library", False, "not-bf16-inference")
    kern = r.kernel
    if own and kern != "library":
        ok = {"ws": lambda: bias is None and linear_ws_eligible(x, weight), "ws128": lambda: bias is None and linear_ws_eligible(x, weight),
              "sm": lambda: linear_sm_eligible(x, weight, bias), "tiled": lambda: True,
              "tiled_halves": lambda: bias is None and x.dim() == 3 and linear_eligible(x, weight[:n // 2])}[kern]()
        if not ok:
            routing.REFUSED.append((role, tokens, n, k, kern))
            del routing.REFUSED[:-64]
            kern = "library"
    fuse = residual is not None and r.fuse_add and kern in ("sm", "tiled") and not torch.is_grad_enabled() and gated_residual_eligible(x, residual, gate)
    if kern == "library":
        y = linear_train(x, weight, bias)
    elif kern == "tiled_halves":
        y = torch.empty(*x.shape[:-1], n, device=x.device, dtype=x.dtype)
        o2 = y.view(-1, n)
        linear(x, weight[:n // 2], out=o2[:, :n // 2])
        linear(x, weight[n // 2:], out=o2[:, n // 2:])
    elif fuse:
        return linear(x, weight, bias, residual=residual, gate=gate, few_tokens=kern == "sm")
    else:
        y = linear(x, weight, bias, weight_stationary=kern in ("ws", "ws128"), few_tokens=kern == "sm")
    return y if residual is None else torch.addcmul(residual, gate.unsqueeze(1), y)


def fuses_gated_add(role, tokens, n, k):
    """will project(role, ...) carry the gated add in the projection's epilogue for this shape (bf16 inference)?"""
    r = routing.route(role, tokens, n, k)
    return r.fuse_add and r.kernel in ("sm", "tiled")
