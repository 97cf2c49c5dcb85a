"""TEST-ONLY stand-ins for the three HIP entry points, backed by the CPU oracle, so that the host-side logic
of zigma_amd (layouts, row tables, pending-residual flow, module plumbing) can be exercised without a GPU.
Installed by monkeypatching inside tests; never imported by the product."""
import numpy as np
import torch

from oracle import zigma_oracle as zo


def _np(t):
    return None if t is None else t.detach().float().cpu().numpy()


def _split(a, period):
    """(B, D, n * period) -> (B * n, D, period): the independent sequences a reset_period concatenates along seqlen."""
    Bn, D, Lt = a.shape
    return a.reshape(Bn, D, Lt // period, period).transpose(0, 2, 1, 3).reshape(-1, D, period)


def _join(a, Bn, period):
    D = a.shape[1]
    return a.reshape(Bn, -1, D, period).transpose(0, 2, 1, 3).reshape(Bn, D, -1)


def conv_raw(x, weight, bias, silu, *, out=None, x_row_index=None, reset_period=0):
    xs = _np(x)
    if x_row_index is not None:
        xs = xs[:, :, x_row_index.long().cpu().numpy()]
    if reset_period:
        y = _join(zo.causal_conv1d(_split(xs, reset_period), _np(weight), _np(bias), "silu" if silu else None),
                  xs.shape[0], reset_period)
    else:
        y = zo.causal_conv1d(xs, _np(weight), _np(bias), "silu" if silu else None)
    y = torch.from_numpy(np.ascontiguousarray(y)).to(x.dtype)
    if out is None:
        out = torch.empty_like(x)
    out.copy_(y)
    return out


def scan_raw(u, delta, A, B, C, D=None, z=None, delta_bias=None, delta_softplus=False, *, out=None, out_z=None,
             x=None, z_row_index=None, out_row_index=None, want_out=True, checkpoints=None, reset_period=0,
             chunk_len=2048, z_preactivated=False, info=None, dt_x=None, dt_w=None, accumulate=False):
    assert checkpoints is None and not z_preactivated and not accumulate, "stand-in: GPU-only features"
    if delta is None or dt_x is not None:             # ABI 9 / 10: dt_proj inside the scan (delta absent, or the split's workspace); (B, L, >= R) rows x (D, R) -> (B, D, L)
        delta = torch.einsum("blr,dr->bdl", dt_x[:, :, :dt_w.shape[1]].float(), dt_w.float()).to(u.dtype)
    if info is not None:
        info[:] = [2, 0]
    zs = _np(z)
    if zs is not None and z_row_index is not None:
        zs = zs[:, :, z_row_index.long().cpu().numpy()]
    if reset_period:                                  # independent sequences of reset_period steps along seqlen
        Bn, P = u.shape[0], reset_period
        bc = lambda M: _split(_np(M)[:, 0], P)[:, None]                     # (B, 1, N, L) -> (B * n, 1, N, P)
        y, last = zo.selective_scan(_split(_np(u), P), _split(_np(delta), P), _np(A), bc(B), bc(C), _np(D), None,
                                    _np(delta_bias), delta_softplus, return_last_state=True)
        y = np.ascontiguousarray(_join(y, Bn, P))
        assert x is None
    else:
        y, last = zo.selective_scan(_np(u), _np(delta), _np(A), _np(B), _np(C), _np(D), None, _np(delta_bias), delta_softplus,
                                    return_last_state=True)
    yz = y * zo.silu(zs) if zs is not None else None

    def place(arr):
        if out_row_index is None:
            return arr
        o = np.empty_like(arr)
        o[:, :, out_row_index.long().cpu().numpy()] = arr
        return o
    if z is not None:
        if out_z is None:
            out_z = torch.empty_like(z)
        out_z.copy_(torch.from_numpy(place(yz)).to(u.dtype))
    if out is None and (want_out or z is None):
        out = torch.empty_like(delta)
    if out is not None:
        out.copy_(torch.from_numpy(place(y)).to(u.dtype))
    if x is not None:
        x.zero_()
        x[:, :, -1, 1::2] = torch.from_numpy(last)
    return out, out_z


def norm_call(x2, weight, bias, residual2, eps, is_rms, residual_dtype, *, rows_per_batch=None, branch=None,
              gate=None, x_out=None, shift=None, scale=None, want_y=True, want_res=None):
    rows, cols = x2.shape
    rpb = rows_per_batch or max(rows, 1)
    rep = lambda m: _np(m).repeat(rpb, axis=0)[:rows]
    xe = _np(x2)
    if branch is not None:
        xe = (xe + rep(gate) * _np(branch)).astype(np.float32)
        xe = _np(torch.from_numpy(xe).to(x2.dtype))
        if x_out is not None:
            x_out.copy_(torch.from_numpy(xe).to(x2.dtype))
    y, res = zo.fused_add_norm(xe, _np(weight), _np(bias), _np(residual2), eps, True, is_rms)
    if residual2 is not None:
        residual_dtype = residual2.dtype
    if want_res is None:
        want_res = residual2 is not None or (residual_dtype is not None and residual_dtype != x2.dtype)
    res_out = torch.from_numpy(res).to(residual_dtype or x2.dtype) if want_res else None
    yt = torch.from_numpy(y).to(x2.dtype)
    y_mod = None
    if shift is not None:
        y_mod = (yt.float() * (1 + torch.from_numpy(rep(scale))) + torch.from_numpy(rep(shift))).to(x2.dtype)
    return (yt if want_y else None), res_out, y_mod


def install(monkeypatch):
    from zigma_amd import causal_conv1d_interface as cci
    from zigma_amd import layernorm as ln
    from zigma_amd import selective_scan_interface as ssi
    monkeypatch.setattr(cci, "causal_conv1d_raw", conv_raw)
    monkeypatch.setattr(ssi, "causal_conv1d_raw", conv_raw)
    monkeypatch.setattr(ssi, "scan_raw", scan_raw)
    monkeypatch.setattr(ln, "_norm_call", norm_call)
