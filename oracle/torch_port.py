"""Torch-CPU restatement of the reference's pure-CPU path — TEST / BASELINE INFRASTRUCTURE, not product code.

Only tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg may import this package (see oracle/zigma_oracle.py).

Why it exists next to the numpy oracle: the north star asks for "the reference's pure-CPU selective_scan_ref path timed on the
host cores" beside the GPU number.  The reference itself cannot travel to the GPU box (/root/reference is not there), and the
numpy oracle runs its recurrence on one thread with different array kernels.  This file restates the SAME ATen formulation the
reference uses — the (B, D, L, N) `deltaA` / `deltaB_u` tensors built with einsum, the per-step Python loop over L, `torch.stack`,
F.softplus / F.silu / F.conv1d / F.linear — so that its timing is the reference's arithmetic on torch's CPU thread pool:
  selective_scan_ref      dis_mamba/mamba_ssm/ops/selective_scan_interface.py:86-152 (real A, variable B / C of shape (B, N, L))
  causal_conv1d_ref       dis_causal_conv1d/causal_conv1d/causal_conv1d_interface.py:49-65
  mamba_inner_ref         selective_scan_interface.py:636-670
  rms_norm_ref / layer_norm_ref   dis_mamba/mamba_ssm/ops/triton/layernorm.py:19-48
  Mamba zigzag branch     dis_mamba/mamba_ssm/modules/mamba_simple.py:274-298,356-395
  Block / ZigMa forward   model_zigma.py:388-460,911-990 (has_text / class / unconditional; zigzagN / hilbertN / v1)
Pinned: tests/test_oracle_golden.py::test_torch_port_* (lines 123-160) compare it with the golden outputs of the unmodified reference
(tests/golden/*.npz) — the same fixtures that pin the numpy oracle."""
import math

import numpy as np
import torch
import torch.nn.functional as F

from . import zigma_oracle as zo


def selective_scan_ref(u, delta, A, B, C, D=None, z=None, delta_bias=None, delta_softplus=False):
    """u, delta, z: (B, D, L); A: (D, N) real; B, C: (B, N, L).  The reference's formulation, op for op."""
    dtype_in = u.dtype
    u, delta = u.float(), delta.float()
    if delta_bias is not None:
        delta = delta + delta_bias[..., None].float()
    if delta_softplus:
        delta = F.softplus(delta)
    batch, dim, dstate = u.shape[0], A.shape[0], A.shape[1]
    B, C = B.float(), C.float()
    x = A.new_zeros((batch, dim, dstate))
    deltaA = torch.exp(torch.einsum("bdl,dn->bdln", delta, A))              # (B, D, L, N): materialised, as in the reference
    deltaB_u = torch.einsum("bdl,bnl,bdl->bdln", delta, B, u)
    ys = []
    for i in range(u.shape[2]):
        x = deltaA[:, :, i] * x + deltaB_u[:, :, i]
        ys.append(torch.einsum("bdn,bn->bd", x, C[:, :, i]))
    y = torch.stack(ys, dim=2)
    out = y if D is None else y + u * D[:, None]
    if z is not None:
        out = out * F.silu(z)
    return out.to(dtype_in)


def causal_conv1d_ref(x, weight, bias=None, activation=None):
    """x: (B, D, L); weight: (D, W) — depthwise conv with left padding W - 1, cut to L, optional SiLU."""
    dtype_in = x.dtype
    x = x.to(weight.dtype)
    L, W = x.shape[-1], weight.shape[1]
    out = F.conv1d(x, weight.unsqueeze(1), bias, padding=W - 1, groups=weight.shape[0])[..., :L]
    return (out if activation is None else F.silu(out)).to(dtype_in)


def mamba_inner_ref(xz, conv_w, conv_b, x_proj_w, dt_proj_w, out_proj_w, out_proj_b, A, D, delta_bias):
    """xz: (B, 2 Di, L) -> (B, L, E): conv + SiLU, x_proj, dt_proj, scan (softplus, gated by z), out_proj."""
    L = xz.shape[-1]
    R, N = dt_proj_w.shape[1], A.shape[1]
    x, z = xz.chunk(2, dim=1)
    x = causal_conv1d_ref(x, conv_w.reshape(conv_w.shape[0], -1), conv_b, "silu")
    x_dbl = F.linear(x.transpose(1, 2).reshape(-1, x.shape[1]), x_proj_w)               # (B L, R + 2N)
    delta = (dt_proj_w @ x_dbl[:, :R].t()).reshape(dt_proj_w.shape[0], -1, L).transpose(0, 1)      # (B, Di, L)
    Bm = x_dbl[:, R:R + N].reshape(-1, L, N).transpose(1, 2).contiguous()
    Cm = x_dbl[:, R + N:].reshape(-1, L, N).transpose(1, 2).contiguous()
    y = selective_scan_ref(x, delta, A, Bm, Cm, D.float(), z=z, delta_bias=delta_bias.float(), delta_softplus=True)
    return F.linear(y.transpose(1, 2), out_proj_w, out_proj_b)


def _norm(x, weight, residual, eps, rms):
    """residual add + RMSNorm / LayerNorm in fp32 (rms_norm_ref / layer_norm_ref); returns (y, residual_out)"""
    xf = x.float() if residual is None else x.float() + residual.float()
    if rms:
        y = xf * torch.rsqrt(xf.square().mean(-1, keepdim=True) + eps)
        y = y if weight is None else y * weight.float()
    else:
        y = F.layer_norm(xf, xf.shape[-1:], weight=None if weight is None else weight.float(), eps=eps)
    return y.to(x.dtype), xf


class ZigMaTorchPort:
    """eval-mode forward of the reference ZigMa from a state dict, for the scan types the hot path benchmarks
    (zigzagN / hilbertN / v1; has_text / num_classes / unconditional).  Tables come from the (golden-pinned) numpy oracle."""

    def __init__(self, state, cfg):
        self.w = {k: torch.as_tensor(np.asarray(v)).float() for k, v in state.items()}
        self.cfg = dict(patch_size=1, has_text=False, num_classes=-1, norm_epsilon=1e-5, scan_type="v2", use_pe=0)
        self.cfg.update(cfg)
        c = self.cfg
        st, side = c["scan_type"], c["img_dim"] // c["patch_size"]
        self.paths = self.paths_rev = None
        if st.startswith(("zigzagN", "hilbertN")):
            k = int(st.replace("zigzagN", "").replace("hilbertN", ""))
            tabs = (zo.zigzag_paths if st.startswith("zigzagN") else zo.hilbert_paths)(side)[:k] * c["depth"]
            self.paths = [torch.as_tensor(np.asarray(p, dtype=np.int64)) for p in tabs]
            self.paths_rev = [torch.as_tensor(np.asarray(zo.reverse_permutation(p), dtype=np.int64)) for p in tabs]
        elif st != "v1":
            raise NotImplementedError(f"torch port: scan_type {st}")

    def mixer(self, i, x):
        w, p = self.w, f"blocks.{i}.mixer."
        xz = F.linear(x, w[p + "in_proj.weight"]).transpose(1, 2)                        # (B, 2 Di, L)
        if self.paths is not None:
            xz = xz[:, :, self.paths[i]]
        o = mamba_inner_ref(xz, w[p + "conv1d.weight"], w[p + "conv1d.bias"], w[p + "x_proj.weight"], w[p + "dt_proj.weight"],
                            w[p + "out_proj.weight"], w.get(p + "out_proj.bias"), -torch.exp(w[p + "A_log"]), w[p + "D"],
                            w[p + "dt_proj.bias"])
        return o if self.paths is None else o[:, self.paths_rev[i], :]

    def cross_attention(self, i, x, text, heads=8):
        w, p = self.w, f"blocks.{i}.msa."
        q, k, v = F.linear(x, w[p + "to_q.weight"]), F.linear(text, w[p + "to_k.weight"]), F.linear(text, w[p + "to_v.weight"])
        B, L, inner = q.shape
        sp = lambda t: t.reshape(B, -1, heads, inner // heads).transpose(1, 2)
        o = F.scaled_dot_product_attention(sp(q), sp(k), sp(v))                          # (CPU math path, model_zigma.py:123)
        return F.linear(o.transpose(1, 2).reshape(B, L, inner), w[p + "to_out.0.weight"], w[p + "to_out.0.bias"])

    def forward(self, x, t, y=None):
        c, w = self.cfg, self.w
        x, t = torch.as_tensor(x).float(), torch.as_tensor(t).float()
        p, E = c["patch_size"], c["embed_dim"]
        B, Cin, H, W = x.shape
        g = x.reshape(B, Cin, H // p, p, W // p, p).permute(0, 2, 4, 1, 3, 5).reshape(B, -1, Cin * p * p)
        h = F.linear(g, w["x_embedder.proj.weight"].reshape(E, -1), w["x_embedder.proj.bias"])
        half = 128
        freqs = torch.exp(-math.log(10000.0) * torch.arange(half, dtype=torch.float32) / half)
        args = (t * 1000.0)[:, None] * freqs[None]
        emb = torch.cat([torch.cos(args), torch.sin(args)], dim=-1)
        tt = F.linear(F.silu(F.linear(emb, w["t_embedder.mlp.0.weight"], w["t_embedder.mlp.0.bias"])),
                      w["t_embedder.mlp.2.weight"], w["t_embedder.mlp.2.bias"])
        text = None
        if c["has_text"]:
            text = F.linear(torch.as_tensor(y).float(), w["y_embedder.weight"], w["y_embedder.bias"])
            cond = tt + text.mean(1)
        elif c["num_classes"] > 0:
            cond = tt + w["y_embedder.embedding_table.weight"][torch.as_tensor(y).long()]
        else:
            cond = tt
        if c["use_pe"] in (1, 2):
            h = h + w["pos_embed"]
        res = None
        for i in range(c["depth"]):
            n, res = _norm(h, w[f"blocks.{i}.norm.weight"], res, c["norm_epsilon"], True)
            mod = F.linear(F.silu(cond), w[f"blocks.{i}.adaLN_modulation.1.weight"], w[f"blocks.{i}.adaLN_modulation.1.bias"])
            ch = mod.chunk(mod.shape[1] // E, dim=1)
            h = n + ch[2][:, None] * self.mixer(i, n * (1 + ch[1][:, None]) + ch[0][:, None])
            if c["has_text"]:
                ln, _ = _norm(h, None, None, 1e-6, False)
                h = h + ch[5][:, None] * self.cross_attention(i, ln * (1 + ch[4][:, None]) + ch[3][:, None], text)
        h, _ = _norm(h, w["norm_f.weight"], res, c["norm_epsilon"], True)
        h, _ = _norm(h, None, None, 1e-6, False)
        h = F.linear(h, w["final_layer.linear.weight"], w["final_layer.linear.bias"])
        s = int(round(math.sqrt(h.shape[1])))
        return h.reshape(B, s, s, p, p, Cin).permute(0, 5, 1, 3, 2, 4).reshape(B, Cin, s * p, s * p)
