"""Module objects with the names and entry points of the reference's two native extensions, backed by
libzigma_hip.so.  `install()` registers them in sys.modules so that the reference's own Python
(`dis_mamba/mamba_ssm/ops/selective_scan_interface.py:9-11`, `causal_conv1d_interface.py:7`) imports them
unchanged:

    selective_scan_cuda.fwd(u, delta, A, B, C, D_, z_, delta_bias_, delta_softplus) -> [out, x(, out_z)]
    causal_conv1d_cuda.causal_conv1d_fwd(x, weight, bias_, silu_activation) -> out

    selective_scan_cuda.bwd(u, delta, A, B, C, D_, z_, delta_bias_, dout, x_, out_, dz_, delta_softplus, recompute_out_z)
        -> [du, ddelta, dA, dB, dC, dD, ddelta_bias(, dz(, out_z))]                       (selective_scan.cpp:338-492)
    causal_conv1d_cuda.causal_conv1d_bwd(x, weight, bias_, dout, dx_, silu_activation) -> [dx, dweight, dbias]
                                                                                          (causal_conv1d.cpp:191-283)

The backward kernels are token-major (channel contiguous); the reference hands (batch, dim, seqlen) tensors, so these
two entry points pay one transposing copy per activation operand — the fused token-major path (`MambaInnerTokFn`)
does not.  `causal_conv1d_update` (recurrent decoding) is out of scope and raises NotImplementedError.
"""
import sys
import types

import torch.nn.functional as F

from .causal_conv1d_interface import causal_conv1d_fwd, conv_bwd_tok
from .selective_scan_interface import scan_bwd_tok, selective_scan_cuda_fwd


def _later(name):
    def fn(*a, **k):
        raise NotImplementedError(f"zigma_amd: {name} is not built yet (out of scope, SURVEY.md §8f)")
    return fn


def _tok(t):
    return None if t is None else t.transpose(1, 2).contiguous()


def selective_scan_cuda_bwd(u, delta, A, B, C, D_, z_, delta_bias_, dout, x_, out_, dz_, delta_softplus, recompute_out_z):
    """Drop-in for `selective_scan_cuda.bwd`.  x_ (the chunk carries) is not needed: the kernel writes its own
    checkpoints.  Constant or grouped B/C are not supported by the backward kernel (ZigMa never produces them)."""
    if B.dim() != 4 or C.dim() != 4 or B.shape[1] != 1 or C.shape[1] != 1:
        raise NotImplementedError("zigma_amd: selective_scan_cuda.bwd supports variable B/C with one group")
    if A.is_complex():
        raise NotImplementedError("zigma_amd: complex A is out of scope")
    if z_ is not None and out_ is None:
        raise RuntimeError("selective_scan_cuda.bwd: the gated backward needs `out`")
    du, ddelta, dA, dB, dC, dD, dz, dbias = scan_bwd_tok(
        _tok(u), _tok(delta), A.float().contiguous(), _tok(B[:, 0]), _tok(C[:, 0]),
        None if D_ is None else D_.float().contiguous(), _tok(z_), None if delta_bias_ is None else delta_bias_.float().contiguous(),
        _tok(dout), _tok(out_), bool(delta_softplus))
    # dB / dC leave in the dtype of B / C like the reference's entry point (selective_scan.cpp:488)
    res = [du.transpose(1, 2), ddelta.transpose(1, 2), dA, dB.transpose(1, 2).unsqueeze(1).to(B.dtype),
           dC.transpose(1, 2).unsqueeze(1).to(C.dtype), dD, dbias]
    if z_ is not None:
        dz = dz.transpose(1, 2)
        if dz_ is not None:
            dz_.copy_(dz)
            dz = dz_
        res.append(dz)
        if recompute_out_z:
            res.append((out_.float() * F.silu(z_.float())).to(out_.dtype))
    return res


def causal_conv1d_cuda_bwd(x, weight, bias_, dout, dx_, silu_activation):
    """Drop-in for `causal_conv1d_cuda.causal_conv1d_bwd` -> [dx, dweight, dbias]."""
    dx, dw, db = conv_bwd_tok(_tok(x), weight, bias_, _tok(dout), bool(silu_activation))
    dx = dx.transpose(1, 2)
    if dx_ is not None:
        dx_.copy_(dx)
        dx = dx_
    return [dx, dw.to(weight.dtype), None if db is None else db.to(bias_.dtype)]


selective_scan_cuda = types.ModuleType("selective_scan_cuda")
selective_scan_cuda.fwd = selective_scan_cuda_fwd
selective_scan_cuda.bwd = selective_scan_cuda_bwd

causal_conv1d_cuda = types.ModuleType("causal_conv1d_cuda")
causal_conv1d_cuda.causal_conv1d_fwd = causal_conv1d_fwd
causal_conv1d_cuda.causal_conv1d_bwd = causal_conv1d_cuda_bwd
causal_conv1d_cuda.causal_conv1d_update = _later("causal_conv1d_cuda.causal_conv1d_update")


def install():
    sys.modules.setdefault("selective_scan_cuda", selective_scan_cuda)
    sys.modules.setdefault("causal_conv1d_cuda", causal_conv1d_cuda)
    return selective_scan_cuda, causal_conv1d_cuda
