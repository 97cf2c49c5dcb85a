"""Deterministic parameter fill shared by the round-2 golden generator (oracle/make_golden_r2.py, run on the reference)
and the tests that rebuild the same model on the HIP path: full-size models (README config: 117 M parameters) cannot
ship their weights as fixtures, so both sides regenerate them from a seed with numpy, key by key in sorted order.

FIXTURE DATA in generator form (kept next to the .npz files it completes; imported by the golden generators through oracle/param_fill.py, by the tests, and by
bench.py's reference check of the timed path — never by the product path).  The values are plausible for a trained ZigMa rather than
the zero-gate default init: S4D-real A with noise, dt biases in softplus^-1([1e-3, 0.1]), O(1) gates."""
import math

import numpy as np
import torch


def _value(key, shape, rng):
    n = lambda s=1.0: (rng.standard_normal(shape) * s).astype(np.float32)
    if key.endswith(("A_log", "A_b_log")):
        base = np.log(np.arange(1, shape[1] + 1, dtype=np.float32))[None, :].repeat(shape[0], 0)
        return base + n(0.2)
    if key.endswith((".D", ".D_b")):
        return 1.0 + n(0.2)
    if key.endswith(("dt_proj.bias", "dt_proj_b.bias")):
        dt = np.exp(rng.random(shape) * (math.log(0.1) - math.log(1e-3)) + math.log(1e-3))
        return (dt + np.log(-np.expm1(-dt))).astype(np.float32)
    if key.endswith(("norm.weight", "norm_f.weight")):
        return 1.0 + n(0.1)
    if "pos_embed" in key or "temporal_pos_embedding" in key:
        return n(0.02)
    if "conv1d" in key and key.endswith("weight"):
        return n(0.4)
    if "adaLN_modulation" in key:
        return n(0.03) if key.endswith("weight") else n(0.5)
    if key.endswith("weight"):
        fan_in = int(np.prod(shape[1:])) if len(shape) > 1 else 1
        return n(1.0 / math.sqrt(max(fan_in, 1)))
    return n(0.05)           # remaining biases


def fill_state(module, seed):
    """Overwrite every entry of module.state_dict() (sorted by key) from numpy default_rng(seed); values are generated in
    float32 and cast to the entry's dtype.  Returns the module."""
    rng = np.random.default_rng(seed)
    sd = module.state_dict()
    with torch.no_grad():
        for key in sorted(sd):
            v = _value(key, tuple(sd[key].shape), rng)
            sd[key].copy_(torch.from_numpy(np.ascontiguousarray(v)).to(sd[key].dtype))
    return module
