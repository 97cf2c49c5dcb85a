"""Dense projections on the hand-written MFMA kernel (zigma_linear_fwd): in_proj / out_proj of the Mamba mixer and
to_q / to_out of the cross-attention (reference call sites mamba_simple.py:290-294, selective_scan_interface.py:365,
model_zigma.py:104-135, all `F.linear`).  Under autograd the callers go through wgrad.LinearTrainFn, whose forward product and dX use the same kernels (its own eligibility
check runs with autograd off); linear_eligible itself refuses tensors that require grad."""
import torch

from . import _lib

import os

# Which projections run on zigma_linear_fwd.  Measured at the headline shapes (M = 65 536 tokens), own 4-wave kernel (csrc/linear4w.hip)
# vs hipBLASLt in us (profiles/r03_e_linear4w_probe.jsonl, r03_n_linear4w_epilogue_probe.jsonl; in the forward r03_t_bench_kernel_stats.csv):
#   out_proj + gated add 122 vs 116 + the add in the norm kernel;  to_out + bias + gated add 54 (67 in the forward) vs 70;
#   to_q 46-47 vs 44-47 (a tie);  the whole in_proj (N = 2560) 222 vs 200 as ONE launch, 2 x 104-105 as two half-width launches
#   (N = 1280 each; mamba_simple.IN_PROJ_SPLIT, round 4 — the default, +1.8 % on the forward against the library).
#   Round 4, in_proj as ONE launch again: the weight-stationary kernel (csrc/linear_ws.hip; mamba_simple.IN_PROJ_WS) 184-196 us.
#   "auto" (default): every projection of the inference path the own kernel serves at least as fast as the library (to_q, to_out,
#   out_proj with the block's gated add) — plus the in_proj halves, which the caller requests with prefer_own;
#   "all": every eligible projection;  "off": library only.
LINEAR_POLICY = os.environ.get("ZIGMA_LINEAR", "auto")


def routes_to_4w(m, n, k, bias=None):
    """shapes zigma_linear_fwd serves with the one-wave-per-SIMD kernel (csrc/linear4w.hip): the wide epilogue-free projections"""
    return bias is None and m % 256 == 0 and n % 128 == 0 and k % 64 == 0 and k >= 192 and (m // 256) * ((n + 255) // 256) >= 256


FORCE_8W = os.environ.get("ZIGMA_LINEAR_8W", "0") == "1"      # A/B knob of tools/fwd_4w_ab.sh: pin the 8-wave kernel
AUTO_4W_MAX_N = int(os.environ.get("ZIGMA_4W_MAX_N", "1024"))   # "auto": the 4-wave kernel where it at least ties the library (to_q; not in_proj)


from . import _knobs  # noqa: E402
_knobs.apply(globals(), "linear")


def linear_eligible(x, weight, bias=None, fused_epilogue=False, prefer_own=False):
    """policy (LINEAR_POLICY) + limits of zigma_linear_fwd: bf16, k % 64 == 0, n % 128 == 0, tokens % 8 == 0, aligned contiguous
    rows, no autograd"""
    if not (LINEAR_POLICY != "off" and x.is_cuda and x.dtype == torch.bfloat16 and weight.dtype == torch.bfloat16):
        return False
    m_, (n_, k_) = x.numel() // max(x.shape[-1], 1), weight.shape
    if LINEAR_POLICY == "auto" and bias is None and not fused_epilogue and not prefer_own and not (routes_to_4w(m_, n_, k_) and n_ <= AUTO_4W_MAX_N):
        return False        # (auto: projections with an epilogue the library cannot fuse, and the wide ones the 4-wave kernel takes)
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
    if bias is not None or not linear_eligible(x, weight, None, prefer_own=True):
        return False
    n, k = weight.shape
    m = x.numel() // k
    pw = 256 if k in (512, 640) else 128 if k in (1280, 1536) else 0
    return pw > 0 and n % pw == 0 and n <= 8192 and m % 512 == 0 and m // 512 >= 32 // (n // pw) and x.stride(-2) % 128 == 0


LINEAR_SM_FLAG = 0x8000          # zigma_linear_params_t.flags: ZIGMA_LINEAR_SM (csrc/linear_sm.hip)


def linear_sm_eligible(x, weight, bias=None):
    """limits of the few-token tiled kernel (csrc/linear_sm.hip: tiles of 128 tokens x n / 4 features, one per workgroup — 8192 tokens x 640
    features are exactly 256 tiles): bf16, k % 64 == 0 and k >= 128, tokens % 128 == 0 (tiles of 160, 192 or 128 features), an optional bf16
    bias on an 8-byte boundary — on top of linear_eligible's alignment rules.  The gated residual epilogue: gated_residual_eligible, as for the
    tiled kernels."""
    if not linear_eligible(x, weight, bias, fused_epilogue=True, prefer_own=True):
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
    P.m, P.n, P.k, P.dtype, P.flags = x2.shape[0], n, k, _lib.dtype_id(x), int(_probe_flags) | (LINEAR_WS_FLAG if weight_stationary else LINEAR_SM_FLAG if few_tokens else (0x2000 if FORCE_8W else 0))
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
