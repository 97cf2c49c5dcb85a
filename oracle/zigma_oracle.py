"""CPU ORACLE for the ZigMa denoiser-forward / ODE-sampling hot path.

TEST INFRASTRUCTURE — NOT PRODUCT CODE.  Only `tests/`, `__graft_entry__.smoke()` and
`bench.py`'s `cpu_baseline` leg may import this module, and there only as the checker.
The product package (`zigma_amd/`) never imports anything under `oracle/`.

This is a from-scratch numpy restatement of the reference algorithms (CompVis/zigma);
every function cites the reference file:line it follows.  It is PINNED against the
reference itself: `oracle/make_golden.py` runs the unmodified reference on CPU (through
`oracle/ref_shim.py`) in the build container and stores input/output vectors under
`tests/golden/`; `tests/test_oracle_golden.py` checks this module against them, against
the seed-0 known answers of the reference's own `test_selective_scan.py` fixture recipe
and against the integer digests of the scan-order tables.
Parity status: op level + model level PINNED (golden vectors from the reference run here);
`torchdiffeq.odeint` (third party, unpinned version, absent from /root/reference) is
restated from its published algorithm — fixed-grid Euler / midpoint / Heun / RK4 and the adaptive
dopri5 controller — and that sampler-trajectory parity is UNPINNED (no reference test or vendored source exists).

All math is done in `dt` (float32 by default, float64 on request).
"""
import math

import numpy as np

# --------------------------------------------------------------------------------------
# small numerics helpers
# --------------------------------------------------------------------------------------


def bf16_round(x):
    """Round-to-nearest-even to bfloat16, returned as float32 (what `.to(bfloat16)` does)."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    u = x.view(np.uint32).astype(np.uint64)
    lsb = (u >> 16) & 1
    r = ((u + 0x7FFF + lsb) >> 16) << 16
    out = r.astype(np.uint32).view(np.float32)
    nan = np.isnan(x)
    if nan.any():
        out = out.copy()
        out[nan] = np.nan
    return out.reshape(x.shape)


def fp16_round(x):
    return np.asarray(x, dtype=np.float32).astype(np.float16).astype(np.float32)


def softplus(x):
    """F.softplus, beta=1, threshold=20 (selective_scan_interface.py:103-104)."""
    x = np.asarray(x)
    safe = np.minimum(x, 20.0)
    return np.where(x > 20.0, x, np.log1p(np.exp(safe))).astype(x.dtype)


def sigmoid(x):
    return 1.0 / (1.0 + np.exp(-x))


def silu(x):
    return x * sigmoid(x)


# --------------------------------------------------------------------------------------
# scan-order tables (integer)  — utils/utils_zigzag.py
# --------------------------------------------------------------------------------------


def reverse_permutation(p):
    """inverse table: rev[p[i]] = i  (utils_zigzag.py:136-141)."""
    p = np.asarray(p, dtype=np.int64)
    rev = np.empty_like(p)
    rev[p] = np.arange(p.size, dtype=np.int64)
    return rev


def zigzag_paths(n):
    """The 8 serpentine orders of an n×n token grid (utils_zigzag.py:144-175).

    Order of the list: for each start corner TL, TR, BL, BR: [row-serpentine, column-serpentine].
    """
    out = []
    for (r0, c0, dr, dc) in ((0, 0, 1, 1), (0, n - 1, 1, -1), (n - 1, 0, -1, 1), (n - 1, n - 1, -1, -1)):
        rows = []
        for i in range(n):          # row i, left-to-right on even i
            for j in range(n):
                col = j if i % 2 == 0 else n - 1 - j
                rows.append((r0 + dr * i) * n + c0 + dc * col)
        cols = []
        for j in range(n):          # column j, top-to-bottom on even j
            for i in range(n):
                row = i if j % 2 == 0 else n - 1 - i
                cols.append((r0 + dr * row) * n + c0 + dc * j)
        out.append(np.array(rows, dtype=np.int64))
        out.append(np.array(cols, dtype=np.int64))
    return out


def _sgn(v):
    return (v > 0) - (v < 0)


def _gilbert_walk(x, y, ax, ay, bx, by, emit):
    """Generalised Hilbert curve, generator form of the published algorithm
    (jakubcerveny/gilbert, BSD-2; the reference vendors its xy->index form at
    utils_zigzag.py:16-120).  Visits every cell of the |a|×|b| rectangle once."""
    w, h = abs(ax + ay), abs(bx + by)
    dax, day, dbx, dby = _sgn(ax), _sgn(ay), _sgn(bx), _sgn(by)
    if h == 1:
        for _ in range(w):
            emit(x, y)
            x, y = x + dax, y + day
        return
    if w == 1:
        for _ in range(h):
            emit(x, y)
            x, y = x + dbx, y + dby
        return
    ax2, ay2, bx2, by2 = ax // 2, ay // 2, bx // 2, by // 2
    w2, h2 = abs(ax2 + ay2), abs(bx2 + by2)
    if 2 * w > 3 * h:
        if (w2 % 2) and w > 2:
            ax2, ay2 = ax2 + dax, ay2 + day
        _gilbert_walk(x, y, ax2, ay2, bx, by, emit)
        _gilbert_walk(x + ax2, y + ay2, ax - ax2, ay - ay2, bx, by, emit)
    else:
        if (h2 % 2) and h > 2:
            bx2, by2 = bx2 + dbx, by2 + dby
        _gilbert_walk(x, y, bx2, by2, ax2, ay2, emit)
        _gilbert_walk(x + bx2, y + by2, ax, ay, bx - bx2, by - by2, emit)
        _gilbert_walk(x + (ax - dax) + (bx2 - dbx), y + (ay - day) + (by2 - dby),
                      -bx2, -by2, -(ax - ax2), -(ay - ay2), emit)


def gilbert_order_index(n):
    """order_index[x, y] = position of cell (x, y) on the curve (utils_zigzag.py:123-130)."""
    grid = np.zeros((n, n), dtype=np.int64)
    counter = [0]

    def emit(x, y):
        grid[x, y] = counter[0]
        counter[0] += 1

    _gilbert_walk(0, 0, n, 0, 0, n, emit)   # w >= h branch of gilbert_xy2d (utils_zigzag.py:23-25)
    assert counter[0] == n * n
    return grid


def hilbert_paths(n):
    """8 variants = order-index grid and its transposes / 90° rotations, flattened
    (utils_zigzag.py:285-302).  NB the tables are order-INDEX grids used directly as perms."""
    g = gilbert_order_index(n)
    r90, r180, r270 = np.rot90(g, 1), np.rot90(g, 2), np.rot90(g, 3)
    variants = [g, g.T, r90, r90.T, r180, r180.T, r270, r270.T]
    return [np.ascontiguousarray(v).reshape(-1).astype(np.int64) for v in variants]


# --------------------------------------------------------------------------------------
# ops
# --------------------------------------------------------------------------------------


def causal_conv1d(x, weight, bias=None, activation=None, dt=np.float32):
    """Depthwise causal conv, zero left pad, optional SiLU.
    x (B, D, L); weight (D, W); bias (D,)  — causal_conv1d_interface.py:49-65."""
    x = np.asarray(x, dtype=dt)
    w = np.asarray(weight, dtype=dt)
    _, D, L = x.shape
    W = w.shape[1]
    out = np.zeros_like(x)
    for k in range(W):
        shift = W - 1 - k             # tap k looks `shift` steps into the past
        if shift >= L:
            continue
        out[:, :, shift:] += w[None, :, k, None] * x[:, :, :L - shift]
    if bias is not None:
        out += np.asarray(bias, dtype=dt)[None, :, None]
    if activation in ("silu", "swish"):
        out = silu(out)
    elif activation is not None:
        raise NotImplementedError(activation)
    return out.astype(dt)


def selective_scan(u, delta, A, B, C, D=None, z=None, delta_bias=None, delta_softplus=False,
                   return_last_state=False, dt=np.float32, gate=True):
    """Selective SSM recurrence  h_l = exp(δ_l A) h_{l-1} + δ_l B_l u_l ;  y_l = C_l·h_l + D u_l ;
    out = y · silu(z)      — selective_scan_interface.py:86-152 (real A only).

    u, delta, z: (B, D, L).  A: (D, N).  B, C: (D, N) constant, (B, N, L) or (B, G, N, L).
    gate=False returns the ungated y even if z is given (the extension's `out`, selective_scan.cpp:311).
    """
    u = np.asarray(u, dtype=dt)
    delta = np.asarray(delta, dtype=dt)
    A = np.asarray(A, dtype=dt)
    Bsz, Dm, L = u.shape
    N = A.shape[1]
    if delta_bias is not None:
        delta = delta + np.asarray(delta_bias, dtype=dt)[None, :, None]
    if delta_softplus:
        delta = softplus(delta)

    def expand(M):
        M = np.asarray(M, dtype=dt)
        if M.ndim == 2:                       # (D, N) constant over batch and time
            return None, M
        if M.ndim == 3:                       # (B, N, L)
            M = M[:, None]
        G = M.shape[1]                        # (B, G, N, L) -> per channel group
        return np.repeat(M, Dm // G, axis=1), None

    Bv, Bc = expand(B)
    Cv, Cc = expand(C)
    h = np.zeros((Bsz, Dm, N), dtype=dt)
    y = np.empty((Bsz, Dm, L), dtype=dt)
    for l in range(L):
        dl = delta[:, :, l, None]                           # (B, D, 1)
        a = np.exp(dl * A[None])                            # (B, D, N)
        bl = Bc[None] if Bv is None else Bv[:, :, :, l]
        h = a * h + (dl * u[:, :, l, None]) * bl
        cl = Cc[None] if Cv is None else Cv[:, :, :, l]
        y[:, :, l] = (h * cl).sum(-1)
    out = y if D is None else y + u * np.asarray(D, dtype=dt)[None, :, None]
    if z is not None and gate:
        out = out * silu(np.asarray(z, dtype=dt))
    out = out.astype(dt)
    return (out, h) if return_last_state else out


def fused_add_norm(x, weight, bias=None, residual=None, eps=1e-6, prenorm=False, rms=True,
                   dt=np.float32):
    """residual-add + RMSNorm / LayerNorm, fp32 statistics — layernorm.py:65-120 (kernel),
    :380-422 (host).  Returns y or (y, residual_out)."""
    xf = np.asarray(x, dtype=dt)
    if residual is not None:
        xf = xf + np.asarray(residual, dtype=dt)
    if rms:
        rstd = 1.0 / np.sqrt((xf * xf).mean(-1, keepdims=True) + eps)
        y = xf * rstd
    else:
        mu = xf.mean(-1, keepdims=True)
        xc = xf - mu
        rstd = 1.0 / np.sqrt((xc * xc).mean(-1, keepdims=True) + eps)
        y = xc * rstd
    if weight is not None:
        y = y * np.asarray(weight, dtype=dt)
    if bias is not None:
        y = y + np.asarray(bias, dtype=dt)
    y = y.astype(dt)
    return (y, xf.astype(dt)) if prenorm else y


def mamba_inner(xz, conv_w, conv_b, x_proj_w, dt_proj_w, out_proj_w, out_proj_b, A, D, delta_bias,
                dt=np.float32, out_proj=True):
    """conv+SiLU -> x_proj -> dt_proj -> selective scan (z-gated) -> out_proj
    — selective_scan_interface.py:296-365 / :636-670.   xz: (B, 2*Di, L);  conv_w: (Di, W).
    Returns (B, L, E) (or (B, Di, L) when out_proj=False, :155-224)."""
    xz = np.asarray(xz, dtype=dt)
    Di = xz.shape[1] // 2
    L = xz.shape[2]
    x, z = xz[:, :Di], xz[:, Di:]
    conv_w = np.asarray(conv_w, dtype=dt).reshape(Di, -1)
    u = causal_conv1d(x, conv_w, conv_b, "silu", dt=dt)                 # (B, Di, L)
    R = np.asarray(dt_proj_w).shape[1]
    N = np.asarray(A).shape[1]
    tok = u.transpose(0, 2, 1).reshape(-1, Di)                          # (B*L, Di)
    x_dbl = tok @ np.asarray(x_proj_w, dtype=dt).T                      # (B*L, R+2N)
    delta = (x_dbl[:, :R] @ np.asarray(dt_proj_w, dtype=dt).T)          # (B*L, Di)
    delta = delta.reshape(-1, L, Di).transpose(0, 2, 1)
    Bm = x_dbl[:, R:R + N].reshape(-1, L, N).transpose(0, 2, 1)         # (B, N, L)
    Cm = x_dbl[:, R + N:R + 2 * N].reshape(-1, L, N).transpose(0, 2, 1)
    y = selective_scan(u, delta, A, Bm, Cm, D, z=z, delta_bias=delta_bias, delta_softplus=True, dt=dt)
    if not out_proj:
        return y
    o = y.transpose(0, 2, 1) @ np.asarray(out_proj_w, dtype=dt).T
    if out_proj_b is not None:
        o = o + np.asarray(out_proj_b, dtype=dt)
    return o.astype(dt)


# --------------------------------------------------------------------------------------
# model  (weights come in as a dict of numpy arrays keyed like the reference state_dict)
# --------------------------------------------------------------------------------------


def _linear(x, w, b=None):
    y = x @ w.T
    return y if b is None else y + b


class ZigMaOracle:
    """numpy restatement of model_zigma.ZigMa.forward (model_zigma.py:911-990) in eval mode
    (DropPath = identity).  `cfg` keys mirror the constructor (:549-576)."""

    def __init__(self, state, cfg, dt=np.float32):
        self.dt = dt
        self.w = {k: np.asarray(v, dtype=dt) for k, v in state.items()}
        self.cfg = dict(patch_size=1, has_text=False, num_classes=-1, norm_epsilon=1e-5,
                        scan_type="v2", video_frames=0, tpe=False, use_pe=0, d_state=16, d_conv=4)
        self.cfg.update(cfg)
        c = self.cfg
        self.side = c["img_dim"] // c["patch_size"]
        self.depth = c["depth"]
        self._build_paths()

    # ---- scan-order tables per layer: model_zigma.py:689-794 --------------------------------
    def _build_paths(self):
        c, st, depth = self.cfg, self.cfg["scan_type"], self.depth
        self.paths = self.paths_rev = self.st_order = None
        if st.startswith(("zigzagN", "hilbertN")):
            k = int(st.replace("zigzagN", "").replace("hilbertN", ""))
            tabs = (zigzag_paths if st.startswith("zigzagN") else hilbert_paths)(self.side)[:k]
            assert len(tabs) == k
            tabs = tabs * depth                               # list tiled, indexed by layer_idx
            self.paths = tabs
            self.paths_rev = [reverse_permutation(p) for p in tabs]
        elif st.startswith(("zzvideo_", "video_")):
            order = list(st.split("_", 1)[1]) * depth
            T = c["video_frames"]
            sp = zigzag_paths(self.side) * depth
            sp_rev = [reverse_permutation(p) for p in sp]
            fwd, bwd = np.arange(T, dtype=np.int64), np.arange(T - 1, -1, -1, dtype=np.int64)
            tp, tp_rev = [fwd, bwd] * depth, [bwd, fwd] * depth
            self.paths, self.paths_rev, si, ti = [], [], 0, 0
            for d in range(depth):                            # spatial tables consumed per s-layer
                if order[d] == "s":
                    self.paths.append(sp[si]); self.paths_rev.append(sp_rev[si]); si += 1
                else:
                    self.paths.append(tp[ti]); self.paths_rev.append(tp_rev[ti]); ti += 1
            self.st_order = order
        elif st in ("v1", "v2"):
            pass
        else:
            raise ValueError(st)

    # ---- Mamba mixer: mamba_simple.py:274-444 ---------------------------------------------------
    def mixer(self, i, x):
        w, dt, p = self.w, self.dt, f"blocks.{i}.mixer."
        Bsz, L, _ = x.shape
        xz = (x @ w[p + "in_proj.weight"].T).transpose(0, 2, 1)          # (B, 2Di, L)
        A = -np.exp(w[p + "A_log"])
        args = lambda s="": (w[p + f"conv1d{s}.weight"], w[p + f"conv1d{s}.bias"], w[p + f"x_proj{s}.weight"],
                             w[p + f"dt_proj{s}.weight"])
        st = self.cfg["scan_type"]
        ow, ob = w[p + "out_proj.weight"], w.get(p + "out_proj.bias")
        if st == "v1":
            return mamba_inner(xz, *args(), ow, ob, A, w[p + "D"], w[p + "dt_proj.bias"], dt=dt)
        if st == "v2":                                                   # :304-339
            f = mamba_inner(xz, *args(), None, None, A, w[p + "D"], w[p + "dt_proj.bias"], dt=dt, out_proj=False)
            Ab = -np.exp(w[p + "A_b_log"])
            b = mamba_inner(xz[:, :, ::-1], *args("_b"), None, None, Ab, w[p + "D_b"], w[p + "dt_proj_b.bias"],
                            dt=dt, out_proj=False)
            y = (f + b[:, :, ::-1]).transpose(0, 2, 1)
            return _linear(y, ow, ob).astype(dt)
        perm, rev = self.paths[i], self.paths_rev[i]
        if self.st_order is None:                                        # zigzag / hilbert :356-395
            o = mamba_inner(xz[:, :, perm], *args(), ow, ob, A, w[p + "D"], w[p + "dt_proj.bias"], dt=dt)
            return o[:, rev, :]
        T = self.cfg["video_frames"]                                    # video :396-442
        K = L // T
        C2 = xz.shape[1]
        v = xz.reshape(Bsz, C2, T, K)
        if self.st_order[i] == "s":
            v = v.transpose(0, 2, 1, 3).reshape(Bsz * T, C2, K)          # b c (t k) -> (b t) c k
        else:
            v = v.transpose(0, 3, 1, 2).reshape(Bsz * K, C2, T)          # b c (t k) -> (b k) c t
        o = mamba_inner(v[:, :, perm], *args(), ow, ob, A, w[p + "D"], w[p + "dt_proj.bias"], dt=dt)
        o = o[:, rev, :]
        E = o.shape[-1]
        if self.st_order[i] == "s":
            return o.reshape(Bsz, T * K, E)                              # (b t) k c -> b (t k) c
        return o.reshape(Bsz, K, T, E).transpose(0, 2, 1, 3).reshape(Bsz, T * K, E)

    # ---- cross attention: model_zigma.py:95-135 -----------------------------------------------
    def cross_attention(self, i, x, text, heads=8):
        w, p = self.w, f"blocks.{i}.msa."
        q, k, v = x @ w[p + "to_q.weight"].T, text @ w[p + "to_k.weight"].T, text @ w[p + "to_v.weight"].T
        Bsz, L, inner = q.shape
        hd = inner // heads
        split = lambda t: t.reshape(Bsz, -1, heads, hd).transpose(0, 2, 1, 3)
        q, k, v = split(q), split(k), split(v)
        s = (q @ k.transpose(0, 1, 3, 2)) / math.sqrt(hd)
        s = np.exp(s - s.max(-1, keepdims=True))
        a = s / s.sum(-1, keepdims=True)
        o = (a @ v).transpose(0, 2, 1, 3).reshape(Bsz, L, inner)
        return _linear(o, w[p + "to_out.0.weight"], w[p + "to_out.0.bias"])

    # ---- timestep embedding: model_zigma.py:247-275 -------------------------------------------
    def t_embed(self, t, freq_round=None):
        half = 128
        freqs = np.exp(-math.log(10000.0) * np.arange(half, dtype=self.dt) / half)
        if freq_round is not None:            # reference computes freqs in the MODEL dtype (:259-262)
            freqs = freq_round(freqs).astype(self.dt)
        args = t[:, None].astype(np.float32) * freqs[None].astype(np.float32)
        emb = np.concatenate([np.cos(args), np.sin(args)], -1).astype(self.dt)
        if freq_round is not None:
            emb = freq_round(emb).astype(self.dt)
        w = self.w
        hdn = silu(_linear(emb, w["t_embedder.mlp.0.weight"], w["t_embedder.mlp.0.bias"]))
        return _linear(hdn, w["t_embedder.mlp.2.weight"], w["t_embedder.mlp.2.bias"])

    def patch_embed(self, x):
        """conv2d k=s=p -> (B, L, E) — timm PatchEmbed (third party, restated; unpinned version)."""
        c, w = self.cfg, self.w
        p = c["patch_size"]
        Bsz, Cin, H, W = x.shape
        g = x.reshape(Bsz, Cin, H // p, p, W // p, p).transpose(0, 2, 4, 1, 3, 5).reshape(Bsz, -1, Cin * p * p)
        return g @ w["x_embedder.proj.weight"].reshape(-1, Cin * p * p).T + w["x_embedder.proj.bias"]

    def forward(self, x, t, y=None, freq_round=None, return_intermediates=False):
        c, w, dt = self.cfg, self.w, self.dt
        x = np.asarray(x, dtype=dt)
        T = c["video_frames"]
        if T > 0:                                              # PatchEmbed_Video :66-78
            Bsz = x.shape[0]
            h = self.patch_embed(x.reshape((Bsz * T,) + x.shape[2:])).reshape(Bsz, -1, c["embed_dim"])
        else:
            h = self.patch_embed(x)
        Bsz = h.shape[0]
        tt = self.t_embed(np.asarray(t, dtype=dt) * 1000.0, freq_round)
        text = None
        if c["has_text"]:
            text = _linear(np.asarray(y, dtype=dt), w["y_embedder.weight"], w["y_embedder.bias"])
            cond = tt + text.mean(1)
        elif c["num_classes"] > 0:
            cond = tt + w["y_embedder.embedding_table.weight"][np.asarray(y, dtype=np.int64)]
        else:
            cond = tt
        if c["use_pe"] in (1, 2):
            h = h + w["pos_embed"]
        if T > 0 and c["tpe"]:
            K = h.shape[1] // T
            h = (h.reshape(Bsz, T, K, -1) + w["temporal_pos_embedding"][0][None, :, None, :]).reshape(Bsz, T * K, -1)
        E = c["embed_dim"]
        res, inter = None, []
        for i in range(self.depth):                            # Block.forward :388-460
            n, res = fused_add_norm(h, w[f"blocks.{i}.norm.weight"], None, res, c["norm_epsilon"], True, True, dt)
            mod = _linear(silu(cond), w[f"blocks.{i}.adaLN_modulation.1.weight"],
                          w[f"blocks.{i}.adaLN_modulation.1.bias"])
            ch = [mod[:, j * E:(j + 1) * E][:, None, :] for j in range(mod.shape[1] // E)]
            mix = self.mixer(i, n * (1 + ch[1]) + ch[0])
            h = n + ch[2] * mix
            if c["has_text"]:
                ln = fused_add_norm(h, None, None, None, 1e-6, False, False, dt)
                h = h + ch[5] * self.cross_attention(i, ln * (1 + ch[4]) + ch[3], text)
            if return_intermediates:
                inter.append((mix.copy(), h.copy()))
        h = fused_add_norm(h, w["norm_f.weight"], None, res, c["norm_epsilon"], False, True, dt)
        h = fused_add_norm(h, None, None, None, 1e-6, False, False, dt)          # FinalLayer :329-337
        h = _linear(h, w["final_layer.linear.weight"], w["final_layer.linear.bias"])
        p, co = c["patch_size"], c["in_channels"]
        if T > 0:                                              # unpatchify_video :889-902
            s = int(round(math.sqrt(h.shape[1] // T)))
            out = h.reshape(Bsz, T, s, s, p, p, co).transpose(0, 1, 6, 2, 4, 3, 5).reshape(Bsz, T, co, s * p, s * p)
        else:                                                  # unpatchify :874-887
            s = int(round(math.sqrt(h.shape[1])))
            out = h.reshape(Bsz, s, s, p, p, co).transpose(0, 5, 1, 3, 2, 4).reshape(Bsz, co, s * p, s * p)
        out = out.astype(dt)
        return (out, inter) if return_intermediates else out


# --------------------------------------------------------------------------------------
# fixed-grid ODE samplers  (torchdiffeq restated from its published algorithm; UNPINNED)
# --------------------------------------------------------------------------------------


def sample_ode_fixed(model_fn, x0, num_steps=50, method="euler", t0=0.0, t1=1.0, dt=np.float32):
    """x' = model(x, t·1_B) on t = linspace(t0, t1, num_steps); returns (num_steps, *x.shape)
    — transport/integrators.py:83-123 with `method` a torchdiffeq fixed-grid solver."""
    ts = np.linspace(t0, t1, num_steps).astype(np.float32)
    x = np.asarray(x0, dtype=dt)
    ones = np.ones(x.shape[0], dtype=np.float32)
    f = lambda tt, xx: np.asarray(model_fn(xx, ones * tt), dtype=dt)
    traj = [x]
    for k in range(num_steps - 1):
        ta, tb = ts[k], ts[k + 1]
        hstep = tb - ta
        if method == "euler":
            x = x + hstep * f(ta, x)
        elif method == "midpoint":
            x = x + hstep * f(ta + 0.5 * hstep, x + 0.5 * hstep * f(ta, x))
        elif method in ("heun", "heun2"):
            k1 = f(ta, x)
            k2 = f(tb, x + hstep * k1)
            x = x + 0.5 * hstep * (k1 + k2)
        elif method == "rk4":                                  # torchdiffeq's 3/8-rule variant
            k1 = f(ta, x)
            k2 = f(ta + hstep / 3, x + hstep * k1 / 3)
            k3 = f(ta + hstep * 2 / 3, x + hstep * (k2 - k1 / 3))
            k4 = f(tb, x + hstep * (k1 - k2 + k3))
            x = x + hstep * (k1 + 3 * (k2 + k3) + k4) / 8
        else:
            raise ValueError(method)
        traj.append(x)
    return np.stack(traj)


def sample_ode_dopri5(model_fn, x0, num_steps=50, rtol=1e-3, atol=1e-6, t0=0.0, t1=1.0, dt=np.float64):
    """Adaptive Dormand-Prince 5(4) as torchdiffeq runs it for the reference's default `sampling_method="dopri5"`
    (config/ode/ode.yaml:2-5, transport/integrators.py:122) — restated from the published algorithm, UNPINNED (the
    package is absent).  Plain Python control flow, every quantity a host number: the independent check for the
    product's on-device controller.  Returns (trajectory at linspace(t0, t1, num_steps), nfe, accepted, rejected).

    Controller: h0 by Hairer's rule with exponent 1/5; error ratio = RMS(err / (atol + rtol max(|y0|, |y1|)));
    accept iff ratio <= 1; h *= min(10, max(0.9 ratio^-1/5, 0.2 — or 1 when accepted with ratio < 1)); steps never
    clamp to output times; outputs by the quartic dense output through (y0, y(t+h/2), y1, f0, f1)."""
    A_ = [[], [1 / 5], [3 / 40, 9 / 40], [44 / 45, -56 / 15, 32 / 9],
          [19372 / 6561, -25360 / 2187, 64448 / 6561, -212 / 729],
          [9017 / 3168, -355 / 33, 46732 / 5247, 49 / 176, -5103 / 18656],
          [35 / 384, 0.0, 500 / 1113, 125 / 192, -2187 / 6784, 11 / 84]]
    C_ = [0.0, 1 / 5, 3 / 10, 4 / 5, 8 / 9, 1.0, 1.0]
    B5 = [35 / 384, 0.0, 500 / 1113, 125 / 192, -2187 / 6784, 11 / 84, 0.0]
    B4 = [1951 / 21600, 0.0, 22642 / 50085, 451 / 720, -12231 / 42400, 649 / 6300, 1 / 60]
    MID = [6025192743 / 30085553152 / 2, 0.0, 51252292925 / 65400821598 / 2, -2691868925 / 45128329728 / 2,
           187940372067 / 1594534317056 / 2, -1776094331 / 19743644256 / 2, 11237099 / 235043384 / 2]
    ts = np.linspace(t0, t1, num_steps).astype(np.float32).astype(np.float64)     # th.linspace is float32
    y = np.asarray(x0, dtype=dt)
    ones = np.ones(y.shape[0], dtype=np.float32)
    f = lambda tt, yy: np.asarray(model_fn(yy, ones * np.float32(tt)), dtype=dt)
    rms = lambda v: float(np.sqrt(np.mean(np.square(v.astype(np.float64)))))
    t = float(ts[0])
    f0 = f(t, y)
    scale = atol + np.abs(y) * rtol
    d0, d1 = rms(y / scale), rms(f0 / scale)
    h0 = 1e-6 if (d0 < 1e-5 or d1 < 1e-5) else 0.01 * d0 / d1
    f1 = f(t + h0, y + h0 * f0)
    d2 = rms((f1 - f0) / scale) / h0
    h1 = max(1e-6, h0 * 1e-3) if (d1 <= 1e-15 and d2 <= 1e-15) else (0.01 / max(d1, d2)) ** (1.0 / 5)
    h = min(100 * h0, h1)
    nfe, acc, rej = 2, 0, 0
    traj, nxt = [y], 1
    while nxt < len(ts):
        ks = [f0]
        for i in range(1, 7):
            yi = y + h * sum(a * k for a, k in zip(A_[i], ks) if a != 0.0)
            ks.append(f(t + C_[i] * h, yi))
        nfe += 6
        y1 = y + h * sum(b * k for b, k in zip(B5, ks) if b != 0.0)
        err = h * sum((b5 - b4) * k for b5, b4, k in zip(B5, B4, ks) if b5 != b4)
        ratio = rms(err / (atol + rtol * np.maximum(np.abs(y), np.abs(y1))))
        if ratio <= 1:
            acc += 1
            ymid = y + h * sum(m * k for m, k in zip(MID, ks) if m != 0.0)
            fn = ks[-1]
            ca = 2 * h * (fn - f0) - 8 * (y1 + y) + 16 * ymid
            cb = h * (5 * f0 - 3 * fn) + 18 * y + 14 * y1 - 32 * ymid
            cc = h * (fn - 4 * f0) - 11 * y - 5 * y1 + 16 * ymid
            while nxt < len(ts) and ts[nxt] <= t + h:
                s = (ts[nxt] - t) / h
                traj.append(y + s * (h * f0) + s ** 2 * cc + s ** 3 * cb + s ** 4 * ca)
                nxt += 1
            t, y, f0 = t + h, y1, fn
        else:
            rej += 1
        if ratio == 0:
            fac = 10.0
        else:
            fac = min(10.0, max(0.9 / ratio ** 0.2, 1.0 if ratio < 1 else 0.2))
        h *= fac
    return np.stack(traj), nfe, acc, rej


# ---------------------------------------------------------------------------------------------------
# Backward restatements.  Same role as the forward ones: checker for the HIP
# backward kernels; pinned against autograd through the reference's pure-torch forward
# (oracle/make_golden_bwd.py -> tests/golden/bwd_*.npz).
# ---------------------------------------------------------------------------------------------------
def selective_scan_bwd(u, delta, A, B, C, D, z, delta_bias, dout, delta_softplus=False, dt=np.float64):
    """Gradients of `selective_scan` (real A, variable B/C of shape (B, N, L) / (B, 1, N, L), or constant (D, N)).

    Math of selective_scan_bwd_kernel.cuh:161-329 (what the reference's CUDA backward computes; its python
    reference gets the same numbers from autograd, test_selective_scan.py:100-149):
      g = dout * silu(z);  dz = dout * y * sigmoid(z) * (1 + z * (1 - sigmoid(z)))
      dh_l = g_l C_l + a_{l+1} dh_{l+1};  dC_l = sum_d g_l h_l;  dB_l = sum_d dh_l * delta_l * u_l
      du_l = g_l D + delta_l * sum_n dh_l B_l;  ddelta_l = sum_n dh_l (B_l u_l + A a_l h_{l-1})
      dA = sum_{b,l} dh_l * delta_l * a_l * h_{l-1};  dD = sum g u;  softplus' = sigmoid(delta_raw) (delta_raw <= 20)
    Returns dict(du, ddelta, dA, dB, dC, dD, dz, ddelta_bias) — entries None where the input was None."""
    u = np.asarray(u, dtype=dt)
    draw = np.asarray(delta, dtype=dt)
    A = np.asarray(A, dtype=dt)
    dout = np.asarray(dout, dtype=dt)
    Bsz, Dm, L = u.shape
    N = A.shape[1]
    if delta_bias is not None:
        draw = draw + np.asarray(delta_bias, dtype=dt)[None, :, None]
    dl_all = softplus(draw) if delta_softplus else draw
    Bv = np.asarray(B, dtype=dt)
    Cv = np.asarray(C, dtype=dt)
    var_B, var_C = Bv.ndim >= 3, Cv.ndim >= 3
    if Bv.ndim == 4:
        assert Bv.shape[1] == 1, "oracle backward: one B/C group"
        Bv = Bv[:, 0]
    if Cv.ndim == 4:
        Cv = Cv[:, 0]
    # forward, keeping every state
    H = np.zeros((L + 1, Bsz, Dm, N), dtype=dt)
    a_all = np.empty((L, Bsz, Dm, N), dtype=dt)
    y = np.empty((Bsz, Dm, L), dtype=dt)
    for l in range(L):
        dl = dl_all[:, :, l, None]
        a_all[l] = np.exp(dl * A[None])
        bl = Bv[:, None, :, l] if var_B else Bv[None]
        H[l + 1] = a_all[l] * H[l] + dl * u[:, :, l, None] * bl
        cl = Cv[:, None, :, l] if var_C else Cv[None]
        y[:, :, l] = (H[l + 1] * cl).sum(-1)
    if D is not None:
        y = y + u * np.asarray(D, dtype=dt)[None, :, None]
    if z is not None:
        zf = np.asarray(z, dtype=dt)
        sg = sigmoid(zf)
        g = dout * zf * sg
        dz = dout * y * sg * (1.0 + zf * (1.0 - sg))
    else:
        g, dz = dout, None
    du = np.zeros_like(u)
    ddl = np.zeros_like(u)
    dA = np.zeros((Dm, N), dtype=dt)
    dB = np.zeros_like(Bv)
    dC = np.zeros_like(Cv)
    adh = np.zeros((Bsz, Dm, N), dtype=dt)
    for l in range(L - 1, -1, -1):
        dl = dl_all[:, :, l, None]
        bl = Bv[:, None, :, l] if var_B else Bv[None]
        cl = Cv[:, None, :, l] if var_C else Cv[None]
        dh = g[:, :, l, None] * cl + adh
        ahm = a_all[l] * H[l]                                   # a_l h_{l-1}
        if var_C:
            dC[:, :, l] = (g[:, :, l, None] * H[l + 1]).sum(1)
        else:
            dC += (g[:, :, l, None] * H[l + 1]).sum(0)
        if var_B:
            dB[:, :, l] = (dh * dl * u[:, :, l, None]).sum(1)
        else:
            dB += (dh * dl * u[:, :, l, None]).sum(0)
        sp = (dh * bl).sum(-1)
        du[:, :, l] = dl[:, :, 0] * sp
        ddl[:, :, l] = u[:, :, l] * sp + (dh * A[None] * ahm).sum(-1)
        dA += (dh * dl * ahm).sum(0)
        adh = a_all[l] * dh
    dD = None
    if D is not None:
        du = du + g * np.asarray(D, dtype=dt)[None, :, None]
        dD = (g * u).sum((0, 2))
    if delta_softplus:
        ddl = ddl * np.where(draw <= 20.0, sigmoid(draw), 1.0)
    dbias = ddl.sum((0, 2)) if delta_bias is not None else None
    return dict(du=du, ddelta=ddl, dA=dA, dB=dB, dC=dC, dD=dD, dz=dz, ddelta_bias=dbias)


def causal_conv1d_bwd(x, weight, bias, dout, activation=None, dt=np.float64):
    """Gradients of `causal_conv1d` (causal_conv1d_bwd.cu:46-240): x, dout (B, D, L); weight (D, W).
    Returns (dx, dweight, dbias)."""
    x = np.asarray(x, dtype=dt)
    w = np.asarray(weight, dtype=dt)
    dout = np.asarray(dout, dtype=dt)
    _, Dm, L = x.shape
    W = w.shape[1]
    if activation in ("silu", "swish"):
        pre = causal_conv1d(x, w, bias, None, dt=dt)
        sg = sigmoid(pre)
        dout = dout * sg * (1.0 + pre * (1.0 - sg))
    dx = np.zeros_like(x)
    dw = np.zeros_like(w)
    for k in range(W):
        shift = W - 1 - k
        if shift >= L:
            continue
        dx[:, :, :L - shift] += w[None, :, k, None] * dout[:, :, shift:]
        dw[:, k] = (dout[:, :, shift:] * x[:, :, :L - shift]).sum((0, 2))
    db = dout.sum((0, 2)) if bias is not None else None
    return dx, dw, db


def fused_add_norm_bwd(x, weight, bias, residual, dy, dresidual_out=None, eps=1e-6, rms=True, dt=np.float64):
    """Gradients of `fused_add_norm` (layernorm.py:196-377): returns (dx, dweight, dbias, dresidual).
    dresidual_out is the gradient flowing into the second output of the prenorm form.  x and residual enter
    through their sum, so dx == dresidual (the reference returns the same tensor for both when dtypes match)."""
    xf = np.asarray(x, dtype=dt)
    if residual is not None:
        xf = xf + np.asarray(residual, dtype=dt)
    dy = np.asarray(dy, dtype=dt)
    Ncol = xf.shape[-1]
    w = np.ones(Ncol, dtype=dt) if weight is None else np.asarray(weight, dtype=dt)
    if rms:
        rstd = 1.0 / np.sqrt((xf * xf).mean(-1, keepdims=True) + eps)
        xhat = xf * rstd
    else:
        mu = xf.mean(-1, keepdims=True)
        rstd = 1.0 / np.sqrt(((xf - mu) ** 2).mean(-1, keepdims=True) + eps)
        xhat = (xf - mu) * rstd
    wdy = dy * w
    c1 = (xhat * wdy).mean(-1, keepdims=True)
    if rms:
        dxs = (wdy - xhat * c1) * rstd
    else:
        c2 = wdy.mean(-1, keepdims=True)
        dxs = (wdy - (xhat * c1 + c2)) * rstd
    if dresidual_out is not None:
        dxs = dxs + np.asarray(dresidual_out, dtype=dt)
    lead = tuple(range(dy.ndim - 1))
    dw = (dy * xhat).sum(lead) if weight is not None else None
    db = dy.sum(lead) if bias is not None else None
    return dxs, dw, db, (dxs if residual is not None else None)
