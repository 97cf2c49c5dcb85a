"""The small per-forward operators around the blocks on their own HIP kernels (csrc/embed.hip, csrc/skinny_linear.hip): patch
embedding + bias + position table, sinusoidal timestep features, the timestep MLP / adaLN modulation of all blocks (at most 64 rows of
activations against up to 88 MB of weights), final LayerNorm + output projection.

Mirrors timm's PatchEmbed as ZigMa uses it (reference model_zigma.py:608-614,924,939-940), TimestepEmbedder (:232-275), the adaLN
`SiLU + Linear` of every Block (:441,447) and FinalLayer without conditioning (:313-337).  bf16 inference only (`*_eligible`); anything
else takes the torch composition in model_zigma.py.  No fallback inside the wrappers: they raise when the limits are not met.
"""
import torch

from . import _lib

BF16 = torch.bfloat16
USE_EMBED_KERNELS = True      # False: the torch composition everywhere (A/B in tests and tools)


def _al(t, n):
    return t.data_ptr() % n == 0


def patch_embed_eligible(x, weight, bias, pos):
    if not (USE_EMBED_KERNELS and x.is_cuda and x.dtype == BF16 and weight.dtype == BF16 and x.dim() == 4 and x.stride(3) == 1):
        return False
    E, C, p, _ = weight.shape
    if not weight.is_contiguous() or not _al(weight, 16) or E % 8 or C * p * p * E * 4 > 65536 or x.shape[1] != C or x.shape[2] % p or x.shape[3] % p:
        return False
    if bias is not None and (bias.dtype != BF16 or not bias.is_contiguous() or not _al(bias, 16)):
        return False
    L = (x.shape[2] // p) * (x.shape[3] // p)
    if pos is not None and (pos.dtype != BF16 or pos.shape[-2:] != (L, E) or pos.stride(-1) != 1 or pos.stride(-2) % 8 or not _al(pos, 16)):
        return False
    return x.shape[0] <= 65535


def patch_embed(x, weight, bias=None, pos=None):
    """x (B, C, H, W), weight (E, C, p, p), bias (E), pos (1, L, E) or (L, E) -> (B, L, E) = bf16(bf16(conv + bias) + pos)"""
    dev = _lib.require_device(x, weight, bias, pos)
    if not patch_embed_eligible(x, weight, bias, pos):
        raise RuntimeError("patch_embed: bf16 (B, C, H, W) input, contiguous (E % 8 == 0, C, p, p) weight of at most 64 KB in fp32")
    E, C, p, _ = weight.shape
    Bsz, _, H, W = x.shape
    L = (H // p) * (W // p)
    out = torch.empty(Bsz, L, E, device=x.device, dtype=x.dtype)
    P = _lib.PatchEmbedParams()
    P.batch, P.in_chans, P.height, P.width, P.patch, P.embed_dim, P.dtype, P.flags = Bsz, C, H, W, p, E, _lib.dtype_id(x), 0
    P.x_batch_stride, P.x_chan_stride, P.x_row_stride = x.stride(0), x.stride(1), x.stride(2)
    P.out_batch_stride, P.out_row_stride = out.stride(0), out.stride(1)
    P.x, P.weight, P.bias, P.out = _lib.ptr(x), _lib.ptr(weight), _lib.ptr(bias), _lib.ptr(out)
    if pos is not None:
        P.pos, P.pos_row_stride = _lib.ptr(pos), pos.stride(-2)
    _lib.call("zigma_patch_embed_fwd", P, dev)
    return out


def timestep_embed_eligible(t, freqs):
    return (USE_EMBED_KERNELS and t.is_cuda and t.dtype == BF16 and freqs.dtype == BF16 and t.dim() == 1 and t.is_contiguous()
            and freqs.dim() == 1 and freqs.is_contiguous() and freqs.device == t.device)


def timestep_embed(t, freqs, dim):
    """t (B,) bf16, freqs (dim // 2,) bf16 -> (B, dim) bf16 = [cos(t f), sin(t f)] (products and functions in fp32)"""
    dev = _lib.require_device(t, freqs)
    if not timestep_embed_eligible(t, freqs) or freqs.shape[0] != dim // 2:
        raise RuntimeError("timestep_embed: contiguous bf16 t (B,) and freqs (dim // 2,)")
    out = torch.empty(t.shape[0], dim, device=t.device, dtype=t.dtype)
    P = _lib.TimestepEmbedParams()
    P.batch, P.dim, P.dtype, P.flags, P.out_row_stride = t.shape[0], dim, _lib.dtype_id(t), 0, out.stride(0)
    P.t, P.freqs, P.out = _lib.ptr(t), _lib.ptr(freqs), _lib.ptr(out)
    _lib.call("zigma_timestep_embed_fwd", P, dev)
    return out


def skinny_linear_eligible(x, weight, bias=None):
    if not (USE_EMBED_KERNELS and x.is_cuda and x.dtype == BF16 and weight.dtype == BF16 and x.dim() == 2 and weight.dim() == 2):
        return False
    m, k = x.shape
    n = weight.shape[0]
    if m > 64 or m == 0 or weight.shape[1] != k or n % 16 or k % 128 or k > 1024 or x.stride(1) != 1 or weight.stride(1) != 1:
        return False
    if x.stride(0) % 8 or weight.stride(0) % 8 or not _al(x, 16) or not _al(weight, 16):
        return False
    if bias is not None and (bias.dtype != BF16 or bias.shape != (n,) or not bias.is_contiguous() or not _al(bias, 8)):
        return False
    return not (torch.is_grad_enabled() and (x.requires_grad or weight.requires_grad or (bias is not None and bias.requires_grad)))


def skinny_linear(x, weight, bias=None, silu=False):
    """(m <= 64, k) @ (n, k)^T + bias, optionally on silu(x) rounded to bf16 (the reference's `SiLU -> Linear` in bf16)"""
    dev = _lib.require_device(x, weight, bias)
    if not skinny_linear_eligible(x, weight, bias):
        raise RuntimeError("skinny_linear: bf16, at most 64 rows, n % 16 == 0, k % 128 == 0, k <= 1024, 16-byte aligned rows, no autograd")
    out = torch.empty(x.shape[0], weight.shape[0], device=x.device, dtype=x.dtype)
    P = _lib.SkinnyParams()
    P.m, P.n, P.k, P.dtype, P.flags = x.shape[0], weight.shape[0], x.shape[1], _lib.dtype_id(x), int(bool(silu))
    P.x_row_stride, P.w_row_stride, P.out_row_stride = x.stride(0), weight.stride(0), out.stride(0)
    P.x, P.w, P.bias, P.out = _lib.ptr(x), _lib.ptr(weight), _lib.ptr(bias), _lib.ptr(out)
    _lib.call("zigma_skinny_linear_fwd", P, dev)
    return out


def final_layer_eligible(x, weight, bias):
    if not (USE_EMBED_KERNELS and x.is_cuda and x.dtype == BF16 and weight.dtype == BF16 and x.stride(-1) == 1 and weight.is_contiguous()):
        return False
    n_out, E = weight.shape
    if x.shape[-1] != E or E % 8 or E > 2048 or n_out > 16 or not x.is_contiguous() or not _al(x, 16) or not _al(weight, 16):
        return False
    if bias is not None and (bias.dtype != BF16 or not bias.is_contiguous()):
        return False
    return not (torch.is_grad_enabled() and (x.requires_grad or weight.requires_grad or (bias is not None and bias.requires_grad)))


def final_layer(x, weight, bias, eps):
    """Linear(LayerNorm(x, no affine, eps)) for a projection to at most 16 features: (..., E) -> (..., n_out)"""
    dev = _lib.require_device(x, weight, bias)
    if not final_layer_eligible(x, weight, bias):
        raise RuntimeError("final_layer: contiguous bf16 (..., E % 8 == 0, E <= 2048) input, (n_out <= 16, E) weight, no autograd")
    n_out, E = weight.shape
    x2 = x.reshape(-1, E)
    out = torch.empty(x2.shape[0], n_out, device=x.device, dtype=x.dtype)
    P = _lib.FinalLayerParams()
    P.rows, P.cols, P.n_out, P.dtype, P.flags, P.eps = x2.shape[0], E, n_out, _lib.dtype_id(x), 0, float(eps)
    P.x_row_stride, P.out_row_stride = x2.stride(0), out.stride(0)
    P.x, P.weight, P.bias, P.out = _lib.ptr(x2), _lib.ptr(weight), _lib.ptr(bias), _lib.ptr(out)
    _lib.call("zigma_final_layer_fwd", P, dev)
    return out.view(*x.shape[:-1], n_out)
