"""Module objects with the names and entry points of the reference's two native extensions, backed by
libzigma_hip.so.  `install()` registers them in sys.modules so that the reference's own Python
(`dis_mamba/mamba_ssm/ops/selective_scan_interface.py:9-11`, `causal_conv1d_interface.py:7`) imports them
unchanged:

    selective_scan_cuda.fwd(u, delta, A, B, C, D_, z_, delta_bias_, delta_softplus) -> [out, x(, out_z)]
    causal_conv1d_cuda.causal_conv1d_fwd(x, weight, bias_, silu_activation) -> out

The backward / update entry points are later scope rows; they raise NotImplementedError.
"""
import sys
import types

from .causal_conv1d_interface import causal_conv1d_fwd
from .selective_scan_interface import selective_scan_cuda_fwd


def _later(name):
    def fn(*a, **k):
        raise NotImplementedError(f"zigma_amd: {name} is not built yet (forward-only scope, SURVEY.md §8f)")
    return fn


selective_scan_cuda = types.ModuleType("selective_scan_cuda")
selective_scan_cuda.fwd = selective_scan_cuda_fwd
selective_scan_cuda.bwd = _later("selective_scan_cuda.bwd")

causal_conv1d_cuda = types.ModuleType("causal_conv1d_cuda")
causal_conv1d_cuda.causal_conv1d_fwd = causal_conv1d_fwd
causal_conv1d_cuda.causal_conv1d_bwd = _later("causal_conv1d_cuda.causal_conv1d_bwd")
causal_conv1d_cuda.causal_conv1d_update = _later("causal_conv1d_cuda.causal_conv1d_update")


def install():
    sys.modules.setdefault("selective_scan_cuda", selective_scan_cuda)
    sys.modules.setdefault("causal_conv1d_cuda", causal_conv1d_cuda)
    return selective_scan_cuda, causal_conv1d_cuda
