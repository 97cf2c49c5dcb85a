"""Mamba mixer with ZigMa's scan_type dispatch, on the token-major HIP path.

Mirrors `Mamba` of the reference (dis_mamba/mamba_ssm/modules/mamba_simple.py:64-268 constructor,
:274-444 forward): same constructor signature, same parameter names and shapes (state_dict compatible),
same scan_type vocabulary.  The forward is re-designed for MI355X: activations stay (B, L, C) from the
in_proj GEMM to the out_proj GEMM; the token reordering of every scan type (zigzag / Hilbert / random
permutation, the reversed sweep of `v2`, the per-frame / per-pixel sequences of the video types) is a
row-index table consumed by the conv and scan kernels, so no `index_select`, `flip`, `cat` or
`rearrange(...).contiguous()` pass exists (reference :362-370, :320-337, :388-394).
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _knobs
from . import routing
from .linear import fuses_gated_add, gated_residual_eligible, linear, linear_ws_eligible, project
from .selective_scan_interface import mamba_inner_tok

NO_COPY_TEMPORAL = True      # video "t" layers on strided views (False: the transposing-copy form; A/B in the tests)
# Which kernel serves in_proj / out_proj at which size is zigma_amd/routing.py (ONE table, with the measurement behind every row).
# the SiLU of the gate in in_proj's epilogue (linear_ws_kernel<.., SL>: z leaves as silu(z)) instead of in the scan's (ZIGMA_SCAN_Z_PREACTIVATED):
# 20 of the scan's 311 VALU instructions per tile-wave move into the GEMM's MFMA gaps; the gate is then rounded to bf16 once more than in the
# reference (selective_scan_fwd_kernel.cuh:293 applies silu in fp32 to the bf16 z).  Measured in round 5: a tie in the forward — off.
GATE_IN_IN_PROJ = False
_knobs.apply(globals(), "mamba_simple")      # ZIGMA_KNOBS="mamba_simple.GATE_IN_IN_PROJ=True,..." (A/B tools)


def _int32_table(t, device):
    if t is None:
        return None
    if not torch.is_tensor(t):
        t = torch.as_tensor(t)
    return t.to(device=device, dtype=torch.int32).contiguous()


class Mamba(nn.Module):
    def __init__(self, d_model, d_state=16, d_conv=4, expand=2, dt_rank="auto", dt_min=0.001, dt_max=0.1,
                 dt_init="random", dt_scale=1.0, dt_init_floor=1e-4, conv_bias=True, bias=False,
                 use_fast_path=True, layer_idx=None, device=None, dtype=None, scan_type="v2", **kwargs):
        factory_kwargs = {"device": device, "dtype": dtype}
        super().__init__()
        self.d_model = d_model
        self.d_state = d_state
        self.d_conv = d_conv
        self.expand = expand
        self.d_inner = int(self.expand * self.d_model)
        self.dt_rank = math.ceil(self.d_model / 16) if dt_rank == "auto" else dt_rank
        self.use_fast_path = use_fast_path
        self.layer_idx = layer_idx
        if scan_type.startswith("zzvideo_"):   # the reference's ZigMa emits zzvideo_*, its Mamba only accepts video_*
            scan_type = "video_" + scan_type[len("zzvideo_"):]
        self.scan_type = scan_type

        self.in_proj = nn.Linear(self.d_model, self.d_inner * 2, bias=bias, **factory_kwargs)
        self.conv1d = nn.Conv1d(self.d_inner, self.d_inner, bias=conv_bias, kernel_size=d_conv, groups=self.d_inner,
                                padding=d_conv - 1, **factory_kwargs)
        self.zigzag_paths = kwargs.get("zigzag_paths", None)
        self.zigzag_paths_reverse = kwargs.get("zigzag_paths_reverse", None)
        self.video_frames = kwargs.get("video_frames", None)
        self.st_order = kwargs.get("st_order", None)
        self.extras = kwargs.get("extras", None)
        self.use_jit = kwargs.get("use_jit", False)
        self.activation = "silu"
        self.act = nn.SiLU()

        self.x_proj = nn.Linear(self.d_inner, self.dt_rank + self.d_state * 2, bias=False, **factory_kwargs)
        self.dt_proj = nn.Linear(self.dt_rank, self.d_inner, bias=True, **factory_kwargs)
        self._init_dt(self.dt_proj, dt_init, dt_scale, dt_min, dt_max, dt_init_floor, factory_kwargs)
        self.A_log = nn.Parameter(self._s4d_real_log(device))
        self.A_log._no_weight_decay = True
        self.D = nn.Parameter(torch.ones(self.d_inner, device=device))
        self.D._no_weight_decay = True

        ok = (scan_type in ("v1", "v2") or scan_type.startswith(("video_", "zigzagN", "hilbertN", "randomN", "parallelN")))
        assert ok, f"Invalid scan_type: {scan_type}"

        if scan_type.startswith("parallelN"):
            # constructor-compatible only: the reference builds these lists but has no forward branch for them
            self.parallel_num = int(scan_type.replace("parallelN", ""))
            A_list, conv_list, x_list, dt_list, D_list = [], [], [], [], []
            for _ in range(self.parallel_num):
                self.A_b_log = nn.Parameter(self._s4d_real_log(device))
                self.A_b_log._no_weight_decay = True
                A_list.append(self.A_b_log)
                self.conv1d_b = nn.Conv1d(self.d_inner, self.d_inner, bias=conv_bias, kernel_size=d_conv,
                                          groups=self.d_inner, padding=d_conv - 1, **factory_kwargs)
                conv_list.append(self.conv1d_b)
                self.x_proj_b = nn.Linear(self.d_inner, self.dt_rank + self.d_state * 2, bias=False, **factory_kwargs)
                x_list.append(self.x_proj_b)
                self.dt_proj_b = nn.Linear(self.dt_rank, self.d_inner, bias=True, **factory_kwargs)
                dt_list.append(self.dt_proj_b)
                self.D_b = nn.Parameter(torch.ones(self.d_inner, device=device))
                self.D_b._no_weight_decay = True
                D_list.append(self.D_b)
            self.A_b_log_list = nn.ParameterList(A_list)
            self.conv1d_b_list = nn.ModuleList(conv_list)
            self.x_proj_b_list = nn.ModuleList(x_list)
            self.dt_proj_b_list = nn.ModuleList(dt_list)
            self.D_b_list = nn.ParameterList(D_list)
        elif scan_type == "v2":
            self.A_b_log = nn.Parameter(self._s4d_real_log(device))
            self.A_b_log._no_weight_decay = True
            self.conv1d_b = nn.Conv1d(self.d_inner, self.d_inner, bias=conv_bias, kernel_size=d_conv,
                                      groups=self.d_inner, padding=d_conv - 1, **factory_kwargs)
            self.x_proj_b = nn.Linear(self.d_inner, self.dt_rank + self.d_state * 2, bias=False, **factory_kwargs)
            self.dt_proj_b = nn.Linear(self.dt_rank, self.d_inner, bias=True, **factory_kwargs)
            self.D_b = nn.Parameter(torch.ones(self.d_inner, device=device))
            self.D_b._no_weight_decay = True

        self.out_proj = nn.Linear(self.d_inner, self.d_model, bias=bias, **factory_kwargs)

        # this layer's row-index table (int32, device); non-persistent so the state_dict matches the reference
        perm = None
        if self.zigzag_paths is not None and layer_idx is not None and not scan_type.startswith("parallelN"):
            perm = _int32_table(self.zigzag_paths[layer_idx], device)
        self.register_buffer("_perm", perm, persistent=False)
        # write-back table: out = out'[:, perm_rev]  <=>  out[inverse(perm_rev)[k]] = out'[k].  Equal to _perm when
        # perm_rev really is perm's inverse (zigzag / hilbert); NOT for the reference's temporal tables, which
        # pair [0..T-1] with [T-1..0] (model_zigma.py:765-772) — kept bit-for-bit.
        out_rows = None
        if perm is not None and self.zigzag_paths_reverse is not None:
            rev = torch.as_tensor(self.zigzag_paths_reverse[layer_idx]).to("cpu", torch.int64)
            inv = torch.empty_like(rev)
            inv[rev] = torch.arange(rev.numel(), dtype=torch.int64)
            out_rows = _int32_table(inv, device)
        self.register_buffer("_out_rows", out_rows, persistent=False)
        self._rev_cache = {}
        self._const_cache = {}
        self._tile_cache = {}

    def _s4d_real_log(self, device):
        A = torch.arange(1, self.d_state + 1, dtype=torch.float32, device=device).repeat(self.d_inner, 1).contiguous()
        return torch.log(A)

    @staticmethod
    def _init_dt(dt_proj, dt_init, dt_scale, dt_min, dt_max, dt_init_floor, factory_kwargs):
        dt_init_std = dt_proj.in_features ** -0.5 * dt_scale
        if dt_init == "constant":
            nn.init.constant_(dt_proj.weight, dt_init_std)
        elif dt_init == "random":
            nn.init.uniform_(dt_proj.weight, -dt_init_std, dt_init_std)
        else:
            raise NotImplementedError
        dt = torch.exp(torch.rand(dt_proj.out_features, **factory_kwargs) * (math.log(dt_max) - math.log(dt_min))
                       + math.log(dt_min)).clamp(min=dt_init_floor)
        inv_dt = dt + torch.log(-torch.expm1(-dt))       # softplus^-1
        with torch.no_grad():
            dt_proj.bias.copy_(inv_dt)
        dt_proj.bias._no_reinit = True

    def forward(self, hidden_states, inference_params=None, residual=None, gate=None):
        """residual (B, L, E) + gate (B, E): returns residual + gate[:, None] * mixer(hidden_states) — the block's gated branch add
        (reference model_zigma.py:441-445), carried by out_proj's epilogue where the routing table says so (linear.project)."""
        y = self._mamba_inner_forward(hidden_states, inference_params)
        return project("out_proj", y, self.out_proj.weight, self.out_proj.bias, residual=residual, gate=gate)

    def out_add_fusable(self, residual, gate):
        """True when forward(..., residual=, gate=) will carry the gated add in out_proj's epilogue (no-grad, bf16, 256-row samples, and a row of the
        routing table that fuses at this size)"""
        lin = self.out_proj
        if torch.is_grad_enabled() or not residual.is_cuda or residual.dtype != torch.bfloat16 or residual.dim() != 3:
            return False
        y = torch.empty(residual.shape[0], residual.shape[1], self.d_inner, device="meta", dtype=residual.dtype)   # shape / dtype stand-in
        tokens = residual.shape[1] * residual.shape[0]
        return (fuses_gated_add("out_proj", tokens, lin.weight.shape[0], self.d_inner)
                and lin.weight.dtype == torch.bfloat16 and self.d_inner % 64 == 0
                and lin.weight.shape[0] % 128 == 0 and gated_residual_eligible(y, residual, gate)
                and (lin.bias is None or lin.bias.dtype == torch.bfloat16))

    def _scan_consts(self, sfx):
        """(A = -exp(A_log), D, dt_bias) in float32, as the reference passes them to the scan
        (mamba_simple.py:298,383-384).  They only change when the parameters do, so they are cached against the
        parameters' version counters instead of being recomputed by three eager kernels per layer per forward."""
        A_log, D, dtb = getattr(self, "A" + sfx + "_log"), getattr(self, "D" + sfx), getattr(self, "dt_proj" + sfx).bias
        key = (A_log._version, D._version, dtb._version, A_log.data_ptr(), D.data_ptr(), dtb.data_ptr())
        hit = self._const_cache.get(sfx)
        if hit is None or hit[0] != key or torch.is_grad_enabled():
            vals = (-torch.exp(A_log.float()), D.float(), dtb.float())
            if torch.is_grad_enabled():
                return vals
            hit = (key, vals)
            self._const_cache[sfx] = hit
        return hit[1]

    def _tiled_tables(self, batch, T):
        """(gather, write-back) time tables of length batch * T: entry b * T + t = b * T + table[t]."""
        key = (batch, T, str(self._perm.device))
        hit = self._tile_cache.get(key)
        if hit is None:
            base = (torch.arange(batch, device=self._perm.device, dtype=torch.int32) * T).repeat_interleave(T)
            out_rows = self._out_rows if self._out_rows is not None else self._perm
            hit = ((base + self._perm.repeat(batch)).contiguous(), (base + out_rows.repeat(batch)).contiguous())
            self._tile_cache[key] = hit
        return hit

    def _reversed_table(self, L, device):
        key = (L, str(device))
        if key not in self._rev_cache:
            self._rev_cache[key] = torch.arange(L - 1, -1, -1, device=device, dtype=torch.int32)
        return self._rev_cache[key]

    def _mamba_inner_forward(self, hidden_states, inference_params=None):
        """hidden_states: (B, L, D) -> the gated scan output (B, L, d_inner), before out_proj."""
        if inference_params is not None:
            raise NotImplementedError("zigma_amd: recurrent decoding is out of scope (ZigMa never passes inference_params)")
        batch, seqlen, _ = hidden_states.shape
        A, Dp, dtb = self._scan_consts("")
        st = self.scan_type
        zact = (GATE_IN_IN_PROJ and not torch.is_grad_enabled() and self.in_proj.bias is None and hidden_states.is_cuda
                and hidden_states.dtype == torch.bfloat16 and (st == "v1" or st.startswith(("zigzagN", "hilbertN", "randomN")))
                and self.d_state == 16 and seqlen % 16 == 0 and self.d_inner % 128 == 0 and batch <= 65535
                and routing.route("in_proj", batch * seqlen, 2 * self.d_inner, hidden_states.shape[-1]).kernel == "ws"      # (256-feature panels only: the
                and linear_ws_eligible(hidden_states, self.in_proj.weight))                                                  # narrow form has no SiLU epilogue)
        if zact:      # in_proj writes (x, silu(z)); the scan (hot kernel: 16-bit, 16 states, whole tiles) multiplies by the gate as it finds it
            xz = linear(hidden_states, self.in_proj.weight, weight_stationary=True, silu_from_col=self.d_inner)
        else:
            xz = project("in_proj", hidden_states, self.in_proj.weight, self.in_proj.bias)      # (B, L, 2*Di) token-major
        fwd = lambda t, perm: mamba_inner_tok(t, self.conv1d.weight, self.conv1d.bias, self.x_proj.weight,
                                               self.dt_proj.weight, A, Dp, dtb,
                                               perm=perm, out_rows=self._out_rows if perm is self._perm else None,
                                               delta_softplus=True, z_preactivated=zact)
        if st == "v1":
            y = fwd(xz, None)
        elif st == "v2":
            A_b, Dp_b, dtb_b = self._scan_consts("_b")
            y = fwd(xz, None)
            if torch.is_grad_enabled() and (xz.requires_grad or self.conv1d_b.weight.requires_grad):
                y = y + mamba_inner_tok(xz, self.conv1d_b.weight, self.conv1d_b.bias, self.x_proj_b.weight, self.dt_proj_b.weight, A_b, Dp_b, dtb_b,
                                        perm=self._reversed_table(seqlen, xz.device), delta_softplus=True)      # both already in token order
            else:       # inference: the reversed sweep ADDS itself to y (in its scan's epilogue where the hot kernel serves the call) — no `out + out_b.flip` pass
                y = mamba_inner_tok(xz, self.conv1d_b.weight, self.conv1d_b.bias, self.x_proj_b.weight, self.dt_proj_b.weight, A_b, Dp_b, dtb_b,
                                    perm=self._reversed_table(seqlen, xz.device), delta_softplus=True, add_to=y)
        elif st.startswith(("zigzagN", "hilbertN", "randomN")):
            if self.extras:
                raise NotImplementedError("extras > 0 is never produced by ZigMa (model_zigma.py:686)")
            y = fwd(xz, self._perm)
        elif st.startswith("video_"):
            T = self.video_frames
            K = seqlen // T
            C2 = xz.shape[-1]
            s_or_t = self.st_order[self.layer_idx]
            if s_or_t == "s":       # b (t k) c -> (b t) k c : a pure view
                y = fwd(xz.view(batch * T, K, C2), self._perm).view(batch, seqlen, -1)
            elif s_or_t == "t" and T % 16 == 0 and xz.numel() < 2 ** 29 and NO_COPY_TEMPORAL:
                # b (t k) c -> scan over t for every (b, k) WITHOUT the two transposing copies: batch = k, sequence =
                # (b, t) with stride K rows, conv window and SSM state restart every T steps (reset_period); the time
                # tables are tiled over b.  Strided views in, strided view out — in both directions: under autograd the
                # backward kernels take the same views and d(xz) comes back as a view of a (b t, k, c) allocation.
                perm_bt, out_bt = self._tiled_tables(batch, T)
                xv = xz.view(batch * T, K, C2).transpose(0, 1)
                if torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in (
                        xz, self.conv1d.weight, self.conv1d.bias, self.x_proj.weight, self.dt_proj.weight, A, Dp, dtb)):
                    y = mamba_inner_tok(xv, self.conv1d.weight, self.conv1d.bias, self.x_proj.weight, self.dt_proj.weight, A, Dp, dtb,
                                        perm=perm_bt, out_rows=out_bt, delta_softplus=True, reset_period=T)
                    y = y.transpose(0, 1).reshape(batch, seqlen, C2 // 2)        # (k, b t, c) view of a (b t, k, c) tensor: no copy
                else:
                    y = torch.empty(batch, seqlen, C2 // 2, device=xz.device, dtype=xz.dtype)
                    mamba_inner_tok(xv, self.conv1d.weight, self.conv1d.bias,
                                    self.x_proj.weight, self.dt_proj.weight, A, Dp, dtb, perm=perm_bt, out_rows=out_bt,
                                    delta_softplus=True, reset_period=T, out=y.view(batch * T, K, C2 // 2).transpose(0, 1))
            elif s_or_t == "t":     # b (t k) c -> (b k) t c : one transposing copy in, one out
                xt = xz.view(batch, T, K, C2).transpose(1, 2).reshape(batch * K, T, C2)
                yt = fwd(xt, self._perm)
                y = yt.view(batch, K, T, -1).transpose(1, 2).reshape(batch, seqlen, -1)
            else:
                raise NotImplementedError
        else:
            raise NotImplementedError
        return y                                                                    # (out_proj: forward())
