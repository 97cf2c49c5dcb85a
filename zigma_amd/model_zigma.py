"""ZigMa denoiser (DiT-style zigzag-Mamba) on the MI355X-native path.

Mirrors the reference's `model_zigma.py`: `ZigMa(...)` constructor signature (:549-576),
`forward(hidden_states, t, y=None)` (:911-990), parameter names and shapes (state_dict compatible with
reference checkpoints), `Block` / `create_block` / `CrossAttention` / `TimestepEmbedder` /
`LabelEmbedder` / `FinalLayer` (:95-135, :232-509).

What is different underneath (sampling / eval forward):
  * activations stay token-major (B, L, C) end to end; the zigzag reordering lives inside the conv / scan
    kernels (zigma_amd.mamba_simple);
  * the elementwise glue between sub-layers — gated residual `x + gate * branch`, residual-stream add,
    RMSNorm / LayerNorm, adaLN `modulate` — is ONE fused HIP launch per sub-layer (zigma_hip.h
    zigma_add_norm_fwd) instead of ~10 eager launches; the gated residual of a sub-layer is carried as a
    pending (base, branch, gate) triple into the next sub-layer's norm kernel;
  * the timestep-frequency table is built once (the reference rebuilds it on the host and copies it to the
    device every forward, :259-262);
  * inputs in a dtype other than the parameter dtype are cast at the boundary and the velocity is returned
    in the input dtype, so an fp32 ODE state can drive a bf16 model.
"""
import math
import os
from functools import partial
from typing import Optional

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F
import torch.utils.checkpoint
from torch import Tensor

from .attention import attention_math, cross_attn, cross_attn_eligible, cross_attn_train
from . import embed as _embed
from .layernorm import RMSNorm, block_norm, glue_bwd_eligible, layer_norm_fn, rms_norm_fn, scale_reduce_bwd
from . import routing
from .linear import linear, linear_eligible, project
from .mamba_simple import Mamba
from .wgrad import linear_train
from .scan_paths import hilbert_path, reverse_permut_np, zigzag_path


def modulate(x, shift, scale):
    return x * (1 + scale.unsqueeze(1)) + shift.unsqueeze(1)


class _ModulateFn(torch.autograd.Function):
    """modulate() for the differentiable path with a hand-written backward: 1 + 3 launches instead of the ~8 of the
    autograd graph of `x * (1 + scale) + shift`."""

    @staticmethod
    def forward(ctx, x, shift, scale):
        ctx.save_for_backward(x, scale)
        return torch.addcmul(shift.unsqueeze(1), x, (1 + scale).unsqueeze(1))

    @staticmethod
    def backward(ctx, dy):
        x, scale = ctx.saved_tensors
        if glue_bwd_eligible(dy, x, scale):            # one pass: dx = dy (1 + scale), dshift = sum_L dy, dscale = sum_L dy x
            dx, dscale, dshift = scale_reduce_bwd(dy, x, scale, s_add=1.0, want_sum=True)
            return dx, dshift, dscale
        dx = dy * (1 + scale).unsqueeze(1)
        return dx, dy.sum(1), (dy * x).sum(1)


class _GatedAddFn(torch.autograd.Function):
    """base + gate[:, None, :] * branch with a hand-written backward."""

    @staticmethod
    def forward(ctx, base, gate, branch):
        ctx.save_for_backward(gate, branch)
        return torch.addcmul(base, gate.unsqueeze(1), branch)

    @staticmethod
    def backward(ctx, dy):
        gate, branch = ctx.saved_tensors
        if glue_bwd_eligible(dy, branch, gate):        # one pass: dbranch = dy gate, dgate = sum_L dy branch
            dbranch, dgate, _ = scale_reduce_bwd(dy, branch, gate)
            return dy, dgate, dbranch
        return dy, (dy * branch).sum(1), dy * gate.unsqueeze(1)


class PatchEmbed(nn.Module):
    """2D image -> patch tokens (timm's PatchEmbed surface: .proj conv weights, .num_patches, .patch_size).
    The strided conv is evaluated as unfold + GEMM (kernel == stride, so the two are the same map)."""

    def __init__(self, img_size=224, patch_size=16, in_chans=3, embed_dim=768, bias=True):
        super().__init__()
        self.img_size = (img_size, img_size)
        self.patch_size = (patch_size, patch_size)
        self.grid_size = (img_size // patch_size, img_size // patch_size)
        self.num_patches = self.grid_size[0] * self.grid_size[1]
        self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=patch_size, stride=patch_size, bias=bias)

    def forward(self, x, pos=None):
        """pos: optional (1, L, E) position table added to the tokens (ZigMa adds it right after, model_zigma.py:939-940): with it the
        bf16 inference path is ONE kernel (zigma_patch_embed_fwd) instead of unfold + K = C p^2 GEMM + bias + add"""
        Bsz, Cin, H, W = x.shape
        p = self.patch_size[0]
        no_grad = not (torch.is_grad_enabled() and (x.requires_grad or self.proj.weight.requires_grad
                                                    or (self.proj.bias is not None and self.proj.bias.requires_grad) or (pos is not None and pos.requires_grad)))
        if no_grad and _embed.patch_embed_eligible(x, self.proj.weight, self.proj.bias, pos):
            return _embed.patch_embed(x, self.proj.weight, self.proj.bias, pos)
        cols = x.reshape(Bsz, Cin, H // p, p, W // p, p).permute(0, 2, 4, 1, 3, 5).reshape(Bsz, -1, Cin * p * p)
        tok = F.linear(cols, self.proj.weight.reshape(self.proj.weight.shape[0], -1), self.proj.bias)
        return tok if pos is None else tok + pos


class PatchEmbed_Video(PatchEmbed):
    def forward(self, x, pos=None):
        Bsz, T = x.shape[:2]
        tok = super().forward(x.reshape((Bsz * T,) + x.shape[2:]))
        tok = tok.reshape(Bsz, -1, tok.shape[-1])
        return tok if pos is None else tok + pos


class CrossAttention(nn.Module):
    def __init__(self, query_dim, context_dim=None, heads=8, dim_head=64, dropout=0.0):
        super().__init__()
        inner_dim = dim_head * heads
        context_dim = query_dim if context_dim is None else context_dim
        self.scale = dim_head ** -0.5
        self.heads = heads
        self.to_q = nn.Linear(query_dim, inner_dim, bias=False)
        self.to_k = nn.Linear(context_dim, inner_dim, bias=False)
        self.to_v = nn.Linear(context_dim, inner_dim, bias=False)
        self.to_out = nn.Sequential(nn.Linear(inner_dim, query_dim), nn.Dropout(dropout))

    def _proj_out(self, o, residual, gate):
        """to_out (+ dropout); with residual / gate the block's gated branch add `residual + gate * to_out(o)` (reference Block,
        model_zigma.py:447-449) — in the projection kernel's epilogue where the routing table says so (zigma_amd/routing.py, role "to_out")"""
        lin = self.to_out[0]
        if residual is None or self.training:
            y = self.to_out[1](project("to_out", o, lin.weight, lin.bias))
            return y if residual is None else torch.addcmul(residual, gate.unsqueeze(1), y)
        return project("to_out", o, lin.weight, lin.bias, residual=residual, gate=gate)

    def forward(self, x, text, mask=None, kv=None, residual=None, gate=None):
        """kv: optional precomputed (to_k(text), to_v(text)), each (B, n_ctx, inner) — ZigMa.forward batches these
        projections of all layers into one GEMM since `text` is the same for every block.
        residual (B, L, E), gate (B, E): return residual + gate * attention(x) instead of attention(x)."""
        Bsz, L, _ = x.shape
        H = self.heads
        k, v = kv[:2] if kv is not None else (self.to_k(text), self.to_v(text))
        q = project("to_q", x, self.to_q.weight, self.to_q.bias)
        if cross_attn_eligible(q, k, v, H):
            # HIP kernel: attention core in one pass (K/V of the head in LDS); under autograd its differentiable form (backward =
            # library GEMMs + ATen elementwise ops).  No fused SDPA anywhere: on ROCm that is an AOT-Triton kernel.
            if torch.is_grad_enabled() and (q.requires_grad or k.requires_grad or v.requires_grad):
                return self._proj_out(cross_attn_train(q, k, v, H, self.scale), residual, gate)
            return self._proj_out(cross_attn(q, k, v, H, self.scale), residual, gate)
        # operands the kernel does not take (fp32 / fp16 models, CPU tensors, long contexts): the same math in torch ops
        return self._proj_out(attention_math(q, k.reshape(Bsz, k.shape[1], -1), v.reshape(Bsz, v.shape[1], -1), H, self.scale), residual, gate)


def drop_path(x, drop_prob: float = 0.0, training: bool = False, scale_by_keep: bool = True):
    if drop_prob == 0.0 or not training:
        return x
    keep = 1 - drop_prob
    mask = x.new_empty((x.shape[0],) + (1,) * (x.ndim - 1)).bernoulli_(keep)
    if keep > 0.0 and scale_by_keep:
        mask.div_(keep)
    return x * mask


class DropPath(nn.Module):
    def __init__(self, drop_prob: float = 0.0, scale_by_keep: bool = True):
        super().__init__()
        self.drop_prob = drop_prob
        self.scale_by_keep = scale_by_keep

    def forward(self, x):
        return drop_path(x, self.drop_prob, self.training, self.scale_by_keep)


class TimestepEmbedder(nn.Module):
    """Embeds scalar timesteps into vector representations."""

    def __init__(self, hidden_size, dtype, frequency_embedding_size=256):
        super().__init__()
        self.mlp = nn.Sequential(nn.Linear(frequency_embedding_size, hidden_size, bias=True), nn.SiLU(),
                                 nn.Linear(hidden_size, hidden_size, bias=True))
        self.dtype = dtype
        self.frequency_embedding_size = frequency_embedding_size
        self.register_buffer("_freqs", self.frequencies(frequency_embedding_size, dtype), persistent=False)

    @staticmethod
    def frequencies(dim, dtype, max_period=10000):
        # computed in the MODEL dtype exactly like the reference does (bf16 rounding included, :259-262)
        half = dim // 2
        return torch.exp(-math.log(max_period) * torch.arange(start=0, end=half, dtype=dtype) / half)

    @staticmethod
    def timestep_embedding(t, dim, dtype, max_period=10000, freqs=None):
        if freqs is None:
            freqs = TimestepEmbedder.frequencies(dim, dtype, max_period).to(device=t.device)
        args = t[:, None].float() * freqs[None]
        embedding = torch.cat([torch.cos(args), torch.sin(args)], dim=-1)
        if dim % 2:
            embedding = torch.cat([embedding, torch.zeros_like(embedding[:, :1])], dim=-1)
        return embedding

    def forward(self, t):
        freqs = self._freqs if self._freqs.dtype == self.dtype else self._freqs.to(self.dtype)
        l1, l2 = self.mlp[0], self.mlp[2]
        if self.frequency_embedding_size % 2 == 0 and _embed.timestep_embed_eligible(t, freqs):
            t_freq = _embed.timestep_embed(t, freqs, self.frequency_embedding_size)       # one kernel instead of ~8 elementwise launches
        else:
            t_freq = self.timestep_embedding(t, self.frequency_embedding_size, dtype=self.dtype, freqs=freqs).to(dtype=self.dtype)
        if _embed.skinny_linear_eligible(t_freq, l1.weight, l1.bias):
            h = _embed.skinny_linear(t_freq, l1.weight, l1.bias)
            if _embed.skinny_linear_eligible(h, l2.weight, l2.bias):
                return _embed.skinny_linear(h, l2.weight, l2.bias, silu=True)                # SiLU folded into the staging of the second product
            return l2(F.silu(h))
        return self.mlp(t_freq)


class LabelEmbedder(nn.Module):
    def __init__(self, num_classes, hidden_size, dropout_prob):
        super().__init__()
        use_cfg_embedding = dropout_prob > 0
        self.embedding_table = nn.Embedding(num_classes + use_cfg_embedding, hidden_size)
        self.num_classes = num_classes
        self.dropout_prob = dropout_prob

    def token_drop(self, labels, force_drop_ids=None):
        if force_drop_ids is None:
            drop_ids = torch.rand(labels.shape[0], device=labels.device) < self.dropout_prob
        else:
            drop_ids = force_drop_ids == 1
        return torch.where(drop_ids, self.num_classes, labels)

    def forward(self, labels, train, force_drop_ids=None):
        if (train and self.dropout_prob > 0) or (force_drop_ids is not None):
            labels = self.token_drop(labels, force_drop_ids)
        return self.embedding_table(labels)


class FinalLayer(nn.Module):
    def __init__(self, hidden_size, patch_size, out_channels, cond=False):
        super().__init__()
        self.norm_final = nn.LayerNorm(hidden_size, elementwise_affine=False, eps=1e-6)
        self.linear = nn.Linear(hidden_size, patch_size * patch_size * out_channels, bias=True)
        if cond:
            self.adaLN_modulation = nn.Sequential(nn.SiLU(), nn.Linear(hidden_size, 2 * hidden_size, bias=True))

    def forward(self, x, c=None):
        if c is None and _embed.final_layer_eligible(x, self.linear.weight, self.linear.bias):
            return _embed.final_layer(x, self.linear.weight, self.linear.bias, self.norm_final.eps)      # LayerNorm + projection in one pass
        x = layer_norm_fn(x, None, None, eps=self.norm_final.eps)
        if c is not None:
            shift, scale = self.adaLN_modulation(c).chunk(2, dim=1)
            x = modulate(x, shift, scale)
        return self.linear(x)


class Pending:
    """A sub-layer output not yet materialised: value = base + gate[:, None] * branch (branch may be None)."""
    __slots__ = ("base", "branch", "gate")

    def __init__(self, base, branch=None, gate=None):
        self.base, self.branch, self.gate = base, branch, gate

    def materialize(self):
        if self.branch is None:
            return self.base
        return torch.addcmul(self.base, self.gate.unsqueeze(1), self.branch)


# out_proj on the own projection kernel WITH the block's gated add `n + gate_msa * mixer(.)` in its epilogue, instead of the library
# GEMM + the add inside the following norm kernel.  Per block (profiles/r02_b_bench_kernel_stats.csv vs r02_d_*): out_proj 127 -> 150 us,
# pre-attention add + norm 68 -> 47, and with to_out's gated add: to_out 63 -> 78, pre-mixer add + norm 119 -> 102 — time moves from
# the HBM-bound norm kernels into the projection epilogues, the forward is 0.1-0.3 % faster.
TEXT_PROJ_OWN = True      # (knob: ZIGMA_KNOBS="model_zigma.TEXT_PROJ_OWN=False")
# y_embedder and the batched K / V projection of all blocks (B x 77 text rows) on zigma_linear_fwd, rows padded to 256


def _padded_own_linear(x, weight, bias):
    """x @ weight.T (+ bias) for a row count the projection kernel's 256-row tiles do not divide (B x 77 text tokens): the rows are
    copied into a zero-padded buffer, the kernel runs on the padded count, the real rows come back as a view.  None when the own
    kernel does not take the operands (then the caller keeps F.linear).  Reference call sites: model_zigma.py:668 (y_embedder),
    :104-112 (to_k / to_v of every block, here one batched product)."""
    if not (x.is_cuda and x.dtype == torch.bfloat16) or (torch.is_grad_enabled() and (x.requires_grad or weight.requires_grad)):
        return None
    k = x.shape[-1]
    m = x.numel() // k
    mp = -(-m // 256) * 256
    xp = x.new_zeros(mp, k)
    xp[:m] = x.reshape(m, k)
    if routing.POLICY == "off" or not linear_eligible(xp, weight, bias):
        return None
    return linear(xp, weight, bias)[:m].view(*x.shape[:-1], weight.shape[0])


FUSE_OUT_PROJ_ADD = True       # (module-level knob for tests / tools; no environment switch)
FUSE_OUT_PROJ_ADD_NO_TEXT = True     # the same for blocks without the attention branch (tools/outproj_notext_ab.py: 14.82 -> 14.78 ms on config 3's model)
from . import _knobs  # noqa: E402
_knobs.apply(globals(), "model_zigma")      # ZIGMA_KNOBS="model_zigma.TO_Q_WS_MAX_TOKENS=0,..." (A/B tools)


class Block(nn.Module):
    def __init__(self, dim, mixer_cls, has_text=False, norm_cls=nn.LayerNorm, fused_add_norm=False,
                 residual_in_fp32=False, drop_path=0.0, skip=False):
        """Add -> Norm -> adaLN-modulated Mixer (-> adaLN-modulated cross-attention), returning
        (hidden_states, residual) like the reference Block (model_zigma.py:340-460)."""
        super().__init__()
        self.residual_in_fp32 = residual_in_fp32
        self.fused_add_norm = fused_add_norm
        self.has_text = has_text
        self.mixer = mixer_cls(dim)
        self.norm = norm_cls(dim)
        self.drop_path = DropPath(drop_path) if drop_path > 0.0 else nn.Identity()
        if self.fused_add_norm:
            assert isinstance(self.norm, (nn.LayerNorm, RMSNorm)), "Only LayerNorm and RMSNorm are supported for fused_add_norm"
        self.skip_linear = nn.Linear(2 * dim, dim) if skip else None
        adaln_num = 3 * 2 if self.has_text else 3
        self.adaLN_modulation = nn.Sequential(nn.SiLU(), nn.Linear(dim, adaln_num * dim, bias=True))
        if self.has_text:
            self.msa = CrossAttention(query_dim=dim, context_dim=dim, heads=8, dim_head=64, dropout=0.0)
            self.norm_msa = nn.LayerNorm(dim, elementwise_affine=False, eps=1e-6)

    def forward_fused(self, pend: Pending, residual, c, text=None, mod=None, kv=None):
        """Hot path.  `pend` is the (unmaterialised) input of this block; returns (Pending, residual).
        mod / kv: this block's adaLN modulation rows and cross-attention K/V when the caller has batched them."""
        E = pend.base.shape[-1]
        if mod is None:
            mod = self.adaLN_modulation(c)                                    # (B, 3E | 6E)
        is_rms = isinstance(self.norm, RMSNorm)
        _, residual, n, xm = block_norm(pend.base, self.norm.weight, self.norm.bias, residual, self.norm.eps, is_rms,
                                        residual_in_fp32=self.residual_in_fp32, branch=pend.branch, gate=pend.gate,
                                        shift=mod[:, 0:E], scale=mod[:, E:2 * E])
        # (with the attention branch the following LayerNorm call shrinks to one read + one write; without it the next block's norm reads one
        # tensor instead of base + branch — round 2's 8-wave kernel lost 0.7-0.9 % there against the library, the 4-wave one wins 0.25 %)
        if FUSE_OUT_PROJ_ADD and self.has_text and self.mixer.out_add_fusable(n, mod[:, 2 * E:3 * E]):
            # n + gate_msa * mixer(xm) in out_proj's epilogue (own projection kernel): the following norm reads one tensor, writes one
            h = self.mixer(xm, residual=n, gate=mod[:, 2 * E:3 * E])
            _, _, _, xa = block_norm(h, None, None, None, self.norm_msa.eps, False, residual_in_fp32=False,
                                     shift=mod[:, 3 * E:4 * E], scale=mod[:, 4 * E:5 * E], want_x=False, want_y=False, want_res_out=False)
            return Pending(self.msa(xa, text=text, mask=None, kv=kv, residual=h, gate=mod[:, 5 * E:6 * E])), residual
        if FUSE_OUT_PROJ_ADD_NO_TEXT and not self.has_text and self.mixer.out_add_fusable(n, mod[:, 2 * E:3 * E]):
            return Pending(self.mixer(xm, residual=n, gate=mod[:, 2 * E:3 * E])), residual
        mix = self.mixer(xm)
        if not self.has_text:
            return Pending(n, mix, mod[:, 2 * E:3 * E]), residual
        h, _, _, xa = block_norm(n, None, None, None, self.norm_msa.eps, False, residual_in_fp32=False, branch=mix,
                                 gate=mod[:, 2 * E:3 * E], shift=mod[:, 3 * E:4 * E], scale=mod[:, 4 * E:5 * E],
                                 want_x=True, want_y=False, want_res_out=False)
        # h + gate_msa * attention(xa): the add rides in to_out's epilogue (one read of att less for the next block's add + norm)
        return Pending(self.msa(xa, text=text, mask=None, kv=kv, residual=h, gate=mod[:, 5 * E:6 * E])), residual

    def forward(self, x: Tensor, residual: Optional[Tensor] = None, c=None, text=None, inference_params=None, skip=None):
        if self.skip_linear is not None:
            x = self.skip_linear(torch.cat([x, skip], dim=-1))
        # stochastic depth exactly where the reference applies it (model_zigma.py:406-437): ONCE, on the incoming
        # branch, and only when a residual stream already exists
        if not self.fused_add_norm:
            residual = x if residual is None else residual + self.drop_path(x)
            n = self.norm(residual.to(dtype=self.norm.weight.dtype))
            if self.residual_in_fp32:
                residual = residual.to(torch.float32)
            mod = self.adaLN_modulation(c).chunk(6 if self.has_text else 3, dim=1)
            h = n + mod[2].unsqueeze(1) * self.mixer(modulate(n, mod[0], mod[1]))
            if self.has_text:
                xa = modulate(layer_norm_fn(h, None, None, eps=self.norm_msa.eps), mod[3], mod[4])
                h = h + mod[5].unsqueeze(1) * self.msa(xa, text=text, mask=None)
            return h, residual
        if torch.is_grad_enabled() and (x.requires_grad or any(q.requires_grad for q in self.parameters())):
            return self.forward_train(x, residual, c, text)
        if residual is not None:
            x = self.drop_path(x)          # identity unless .train() with drop_path > 0 (under no_grad)
        pend, residual = self.forward_fused(Pending(x), residual, c, text)
        return pend.materialize(), residual

    def forward_train(self, x, residual, c, text=None):
        """Differentiable form (the reference's own composition, model_zigma.py:416-458): the add+norm, the Mamba inner
        and the pre-attention LayerNorm are autograd Functions over the HIP forward AND backward kernels; modulate,
        gating, projections and attention are torch ops."""
        fn = rms_norm_fn if isinstance(self.norm, RMSNorm) else layer_norm_fn
        x, residual = fn(x if residual is None else self.drop_path(x), self.norm.weight, self.norm.bias, residual=residual,
                         prenorm=True, residual_in_fp32=self.residual_in_fp32, eps=self.norm.eps)
        mod = self.adaLN_modulation(c).chunk(6 if self.has_text else 3, dim=1)
        x = _GatedAddFn.apply(x, mod[2], self.mixer(_ModulateFn.apply(x, mod[0], mod[1])))
        if self.has_text:
            xa = _ModulateFn.apply(layer_norm_fn(x, None, None, eps=self.norm_msa.eps), mod[3], mod[4])
            x = _GatedAddFn.apply(x, mod[5], self.msa(xa, text=text, mask=None))
        return x, residual


def create_block(d_model, ssm_cfg=None, has_text=False, norm_epsilon=1e-5, drop_path=0.0, rms_norm=False,
                 residual_in_fp32=False, fused_add_norm=False, skip=False, layer_idx=None, device=None, dtype=None,
                 scan_type="none", **block_kwargs):
    if ssm_cfg is None:
        ssm_cfg = {}
    factory_kwargs = {"device": device, "dtype": dtype}
    mixer_cls = partial(Mamba, layer_idx=layer_idx, scan_type=scan_type, **ssm_cfg, **block_kwargs, **factory_kwargs)
    norm_cls = partial(nn.LayerNorm if not rms_norm else RMSNorm, eps=norm_epsilon, **factory_kwargs)
    block = Block(d_model, mixer_cls, has_text=has_text, norm_cls=norm_cls, drop_path=drop_path,
                  fused_add_norm=fused_add_norm, residual_in_fp32=residual_in_fp32, skip=skip)
    block.layer_idx = layer_idx
    return block


def _init_weights(module, n_layer, initializer_range=0.02, rescale_prenorm_residual=True, n_residuals_per_layer=1):
    if isinstance(module, nn.Linear):
        if module.bias is not None and not getattr(module.bias, "_no_reinit", False):
            nn.init.zeros_(module.bias)
    elif isinstance(module, nn.Embedding):
        nn.init.normal_(module.weight, std=initializer_range)
    if rescale_prenorm_residual:
        for name, p in module.named_parameters():
            if name in ["out_proj.weight", "fc2.weight"]:
                nn.init.kaiming_uniform_(p, a=math.sqrt(5))
                with torch.no_grad():
                    p /= math.sqrt(n_residuals_per_layer * n_layer)


def get_1d_sincos_pos_embed_from_grid(embed_dim, pos):
    assert embed_dim % 2 == 0
    omega = 1.0 / 10000 ** (np.arange(embed_dim // 2, dtype=np.float64) / (embed_dim / 2.0))
    out = np.outer(pos.reshape(-1), omega)
    return np.concatenate([np.sin(out), np.cos(out)], axis=1)


def get_2d_sincos_pos_embed(embed_dim, grid_size, cls_token=False, extra_tokens=0):
    """MAE-style fixed 2D sin-cos table: first half of the channels encodes the w coordinate grid,
    second half the h grid (model_zigma.py:1018-1070)."""
    coords = np.arange(grid_size, dtype=np.float32)
    gw, gh = np.meshgrid(coords, coords)           # w varies fastest
    emb = np.concatenate([get_1d_sincos_pos_embed_from_grid(embed_dim // 2, gw),
                          get_1d_sincos_pos_embed_from_grid(embed_dim // 2, gh)], axis=1)
    if cls_token and extra_tokens > 0:
        emb = np.concatenate([np.zeros([extra_tokens, embed_dim]), emb], axis=0)
    return emb


class ZigMa(nn.Module):
    """A DiT-styled Mamba model with ZigZag scan."""

    def __init__(self, in_channels: int, embed_dim: int, depth: int, img_dim: int, patch_size: int = 1,
                 has_text: bool = False, num_classes=-1, drop_path_rate=0.1, n_context_token: int = 0,
                 d_context: int = 0, ssm_cfg=None, norm_epsilon: float = 1e-5, rms_norm: bool = True,
                 fused_add_norm=True, residual_in_fp32=True, initializer_cfg=None, scan_type="v2", video_frames=0,
                 tpe=False, device="cuda", use_pe=0, use_jit=True, m_init=True, use_checkpoint=False,
                 dtype=torch.float32, verbose=False):
        self.factory_kwargs = factory_kwargs = {"device": device, "dtype": dtype}
        super().__init__()
        self.in_channels = in_channels
        self.out_channels = in_channels
        self.patch_size = patch_size
        self.embed_dim = embed_dim
        self.tpe = tpe
        self.residual_in_fp32 = residual_in_fp32
        self.fused_add_norm = fused_add_norm
        self.video_frames = video_frames
        self.use_pe = use_pe
        self.use_checkpoint = use_checkpoint
        self.scan_type = scan_type
        num_patches = (img_dim // patch_size) ** 2

        embed_cls = PatchEmbed if video_frames == 0 else PatchEmbed_Video
        self.x_embedder = embed_cls(img_dim, patch_size, self.in_channels, self.embed_dim, bias=True).to(device).to(dtype)
        self.t_embedder = TimestepEmbedder(self.embed_dim, dtype=dtype).to(device).to(dtype)

        if video_frames < 0:
            raise ValueError("video_frames should be >= 0")
        num_patches_4pe = num_patches if video_frames == 0 else num_patches * video_frames
        if self.use_pe == 1:
            self.pos_embed = nn.Parameter(torch.zeros(1, num_patches_4pe, embed_dim, device=device, dtype=dtype),
                                          requires_grad=False)
        elif self.use_pe == 2:
            self.pos_embed = nn.Parameter(torch.zeros(1, num_patches_4pe, embed_dim, device=device, dtype=dtype))
        elif self.use_pe == 3:
            self.pos_embed_list = [nn.Parameter(torch.zeros(1, num_patches_4pe, embed_dim, device=device, dtype=dtype))] * depth
        elif self.use_pe != 0:
            raise ValueError("use_pe should be 0, 1 or 2")
        if self.tpe:
            self.temporal_pos_embedding = nn.Parameter(torch.zeros(1, video_frames, embed_dim, device=device, dtype=dtype))

        self.n_layer = depth
        self.has_text = has_text
        self.num_classes = num_classes
        if has_text:
            self.y_embedder = nn.Linear(d_context, embed_dim).to(device).to(dtype)
        elif num_classes > 0:
            self.y_embedder = LabelEmbedder(num_classes, hidden_size=embed_dim, dropout_prob=0.0).to(device).to(dtype)

        dpr = [x.item() for x in torch.linspace(0, drop_path_rate, self.n_layer)]
        inter_dpr = [0.0] + dpr
        self.drop_path = DropPath(drop_path_rate) if drop_path_rate > 0.0 else nn.Identity()

        self.extras = 0
        block_kwargs = {"use_jit": use_jit}
        side = int(math.sqrt(num_patches))
        if scan_type.startswith(("zigzagN", "hilbertN", "randomN", "parallelN")):
            if scan_type.startswith("zigzagN"):
                k = int(scan_type.replace("zigzagN", ""))
                zz_paths = zigzag_path(N=side)[:k]
                assert len(zz_paths) == k, f"{len(zz_paths)} != {k}"
            elif scan_type.startswith("parallelN"):
                zz_paths = zigzag_path(N=side)[:8]
            elif scan_type.startswith("hilbertN"):
                k = int(scan_type.replace("hilbertN", ""))
                zz_paths = hilbert_path(N=side)[:k]
                assert len(zz_paths) == k, f"{len(zz_paths)} != {k}"
            else:
                k = int(scan_type.replace("randomN", ""))
                zz_paths = [np.random.permutation(side ** 2) for _ in range(k)]
            zz_paths_rev = [reverse_permut_np(p) for p in zz_paths]
            zz_paths = [torch.from_numpy(np.ascontiguousarray(p)).to(device) for p in zz_paths * depth]
            zz_paths_rev = [torch.from_numpy(p).to(device) for p in zz_paths_rev * depth]
            block_kwargs.update(zigzag_paths=zz_paths, zigzag_paths_reverse=zz_paths_rev, extras=self.extras)
        elif scan_type.startswith("zzvideo_"):
            st_order = list(scan_type.replace("zzvideo_", ""))
            assert len(set(st_order)) == 2
            st_order = st_order * depth
            sp = zigzag_path(N=side)
            sp_rev = [reverse_permut_np(p) for p in sp]
            sp = [torch.from_numpy(np.ascontiguousarray(p)).to(device) for p in sp] * depth
            sp_rev = [torch.from_numpy(p).to(device) for p in sp_rev] * depth
            time_p = torch.arange(video_frames, device=device)
            time_n = torch.arange(video_frames - 1, -1, -1, device=device)
            tp, tp_rev = [time_p, time_n] * depth, [time_n, time_p] * depth
            paths, paths_rev = [], []
            for d in range(depth):       # spatial tables are consumed per s-layer, temporal per t-layer
                if st_order[d] == "s":
                    paths.append(sp.pop(0)); paths_rev.append(sp_rev.pop(0))
                elif st_order[d] == "t":
                    paths.append(tp.pop(0)); paths_rev.append(tp_rev.pop(0))
                else:
                    raise ValueError("st_order should be s or t")
            block_kwargs.update(zigzag_paths=paths, zigzag_paths_reverse=paths_rev, extras=self.extras,
                                video_frames=video_frames, st_order=st_order)
        elif scan_type != "v2":
            raise ValueError("scan_type doesn't match")

        self.blocks = nn.ModuleList([
            create_block(embed_dim, has_text=has_text, ssm_cfg=ssm_cfg, norm_epsilon=norm_epsilon, rms_norm=rms_norm,
                         residual_in_fp32=residual_in_fp32, fused_add_norm=fused_add_norm, layer_idx=i,
                         scan_type=scan_type, drop_path=inter_dpr[i], **block_kwargs, **factory_kwargs).to(device).to(dtype)
            for i in range(self.n_layer)])
        self.final_layer = FinalLayer(self.embed_dim, patch_size, self.out_channels).to(device).to(dtype)
        self.norm_f = (nn.LayerNorm if not rms_norm else RMSNorm)(embed_dim, eps=norm_epsilon, **factory_kwargs)

        self.initialize_weights()
        self.m_init = m_init
        if m_init:
            self.apply(partial(_init_weights, n_layer=depth, **(initializer_cfg if initializer_cfg is not None else {})))

    def initialize_weights(self):
        if self.use_pe == 1:
            pe = get_2d_sincos_pos_embed(self.pos_embed.shape[-1], int(self.x_embedder.num_patches ** 0.5))
            self.pos_embed.data.copy_(torch.from_numpy(pe).float().unsqueeze(0))
        w = self.x_embedder.proj.weight.data
        nn.init.xavier_uniform_(w.view([w.shape[0], -1]))
        nn.init.constant_(self.x_embedder.proj.bias, 0)
        nn.init.normal_(self.t_embedder.mlp[0].weight, std=0.02)
        nn.init.normal_(self.t_embedder.mlp[2].weight, std=0.02)
        for block in self.blocks:
            nn.init.constant_(block.adaLN_modulation[-1].weight, 0)
            nn.init.constant_(block.adaLN_modulation[-1].bias, 0)

    def unpatchify(self, x):
        """x: (N, T, patch_size**2 * C) -> imgs (N, C, H, W)."""
        c, p = self.out_channels, self.x_embedder.patch_size[0]
        h = w = int(x.shape[1] ** 0.5)
        assert h * w == x.shape[1]
        x = x.reshape(x.shape[0], h, w, p, p, c).permute(0, 5, 1, 3, 2, 4)
        return x.reshape(x.shape[0], c, h * p, w * p)

    def unpatchify_video(self, x, video_frames):
        c, p = self.out_channels, self.x_embedder.patch_size[0]
        h = w = int((x.shape[1] // video_frames) ** 0.5)
        assert h * w * video_frames == x.shape[1]
        x = x.reshape(x.shape[0], video_frames, h, w, p, p, c).permute(0, 1, 6, 2, 4, 3, 5)
        return x.reshape(x.shape[0], video_frames, c, h * p, w * p)

    def forward(self, hidden_states, t, y=None):
        """x: (N, C, H, W) [or (N, T, C, H, W)] latents; t: (N,) diffusion times; y: (N,) labels or (N, n_ctx, d_ctx) text."""
        in_dtype = hidden_states.dtype
        pdtype = self.x_embedder.proj.weight.dtype
        pos = self.pos_embed if self.use_pe in (1, 2) else None
        hidden_states = self.x_embedder(hidden_states.to(pdtype), pos=pos)      # (N, T, D), position table added
        _B, _T, _D = hidden_states.shape

        t = (t * 1000.0).to(hidden_states)
        t = self.t_embedder(t)                                                  # (N, D)
        if self.has_text:
            yp = y.to(pdtype)
            ye = _padded_own_linear(yp, self.y_embedder.weight, self.y_embedder.bias) if TEXT_PROJ_OWN else None
            y = ye if ye is not None else self.y_embedder(yp)                   # (B, n_ctx, D)
            c = t + y.mean(dim=1)
        elif self.num_classes > 0:
            c = t + self.y_embedder(y, self.training)
        else:
            c = t

        if self.video_frames > 0 and self.tpe:
            K = _T // self.video_frames
            hidden_states = (hidden_states.view(_B, self.video_frames, K, _D)
                             + self.temporal_pos_embedding.view(1, self.video_frames, 1, _D)).view(_B, _T, _D)

        residual = None
        # autograd recording -> the per-block differentiable composition (HIP forward + backward kernels);
        # otherwise (torch.no_grad(), the sampling path) the fully fused forward
        # (any trainable parameter counts: with frozen embedders and trainable blocks neither hidden_states nor c
        # requires grad, and the raw-launch fused path would silently cut the graph)
        needs_grad = torch.is_grad_enabled() and (hidden_states.requires_grad or c.requires_grad
                                                  or any(p.requires_grad for p in self.parameters()))
        stochastic = self.training and not isinstance(self.drop_path, nn.Identity)   # per-block drop_path is live
        if self.fused_add_norm and not needs_grad and not stochastic and self.use_pe != 3:
            pend = Pending(hidden_states.contiguous())
            mods, kvs = self._batched_conditioning(c, y)
            for i, block in enumerate(self.blocks):
                pend, residual = block.forward_fused(pend, residual, c, y, mod=mods[i], kv=kvs[i] if kvs else None)
            is_rms = isinstance(self.norm_f, RMSNorm)
            _, _, hidden_states, _ = block_norm(pend.base, self.norm_f.weight, self.norm_f.bias, residual,
                                                self.norm_f.eps, is_rms, residual_in_fp32=self.residual_in_fp32,
                                                branch=pend.branch, gate=pend.gate)
        else:
            for layer_idx, block in enumerate(self.blocks):
                if self.use_pe == 3:
                    hidden_states = hidden_states + self.pos_embed_list[layer_idx]
                if self.use_checkpoint and needs_grad:      # activation checkpointing per block (reference :953-956)
                    hidden_states, residual = torch.utils.checkpoint.checkpoint(
                        self.ckpt_wrapper(block), hidden_states, residual, c, y, use_reentrant=False)
                else:
                    hidden_states, residual = block(hidden_states, residual=residual, c=c, text=y)
            if not self.fused_add_norm:
                residual = hidden_states if residual is None else residual + self.drop_path(hidden_states)
                hidden_states = self.norm_f(residual.to(dtype=self.norm_f.weight.dtype))
            else:
                fn = rms_norm_fn if isinstance(self.norm_f, RMSNorm) else layer_norm_fn
                hidden_states = fn(self.drop_path(hidden_states), self.norm_f.weight, self.norm_f.bias,
                                   eps=self.norm_f.eps, residual=residual, prenorm=False,
                                   residual_in_fp32=self.residual_in_fp32)

        hidden_states = self.final_layer(hidden_states)
        if self.video_frames > 0:
            hidden_states = self.unpatchify_video(hidden_states, self.video_frames)
        else:
            hidden_states = self.unpatchify(hidden_states)
        return hidden_states.to(in_dtype)

    def _batched_conditioning(self, c, text):
        """adaLN modulations of ALL blocks in one GEMM, and (has_text) the cross-attention K / V projections of all
        blocks in one GEMM: they depend on (t, y) only, identically for every block (model_zigma.py:441,447-449 and
        :104-107 evaluate them block by block: 3 x depth small GEMMs -> 2).  Stacked weights are cached against the
        parameters' version counters."""
        blocks = self.blocks
        plist = [b.adaLN_modulation[-1].weight for b in blocks] + [b.adaLN_modulation[-1].bias for b in blocks]
        if self.has_text:
            plist += [b.msa.to_k.weight for b in blocks] + [b.msa.to_v.weight for b in blocks]
        key = tuple((p._version, p.data_ptr()) for p in plist)
        cache = getattr(self, "_cond_cache", None)
        if cache is None or cache[0] != key:
            Wm = torch.cat([b.adaLN_modulation[-1].weight for b in blocks], 0)
            bm = torch.cat([b.adaLN_modulation[-1].bias for b in blocks], 0)
            Wkv = torch.cat([torch.cat([b.msa.to_k.weight, b.msa.to_v.weight], 0) for b in blocks], 0) if self.has_text else None
            cache = (key, Wm.detach(), bm.detach(), None if Wkv is None else Wkv.detach())
            self._cond_cache = cache
        _, Wm, bm, Wkv = cache
        n = len(blocks)
        if _embed.skinny_linear_eligible(c, Wm, bm):       # <= 64 samples against n * 6E weight rows: weight-streaming MFMA kernel
            mods_all = _embed.skinny_linear(c, Wm, bm, silu=True)
        else:
            mods_all = F.linear(F.silu(c), Wm, bm)
        mods = mods_all.view(c.shape[0], n, -1).unbind(1)                           # n x (B, 3E | 6E), row pitch n*6E
        kvs = None
        if self.has_text:
            inner = blocks[0].msa.to_k.weight.shape[0]
            kv_own = _padded_own_linear(text, Wkv, None) if TEXT_PROJ_OWN else None
            kv_all = (kv_own if kv_own is not None else F.linear(text, Wkv)).view(text.shape[0], text.shape[1], n, 2, inner)
            kvs = [(kv_all[:, :, i, 0], kv_all[:, :, i, 1]) for i in range(n)]
        return mods, kvs

    def ckpt_wrapper(self, module):
        def ckpt_forward(*inputs):
            return module(*inputs)
        return ckpt_forward

    def forward_with_cfg(self, x, t, y, cfg_scale):
        raise NotImplementedError
