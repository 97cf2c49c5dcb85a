"""Golden vectors for transport.Sampler.sample_sde, produced by the UNMODIFIED reference transport package.

TEST INFRASTRUCTURE ONLY (build container; needs /root/reference).  The reference's integrators.py imports
torchdiffeq (third party, absent): a stub module is registered — the SDE solver does not use it.
    python -m oracle.make_golden_sde   ->  tests/golden/sde_sampler.npz
(`diffusion_form="constant"` is not covered: the reference's Euler step calls th.sqrt on a Python float and raises.)
The toy model is a fixed closed-form function of (x, t), so the trajectories depend only on the reference's
sampler logic and on torch's CPU RNG stream (the reference draws its noise with th.randn on the CPU).
"""
import importlib
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def toy_model(x, t, **kw):
    tt = t.view(-1, *([1] * (x.dim() - 1)))
    return torch.tanh(x) * (0.3 + tt) - 0.5 * x


CASES = [  # (path, prediction, sampler, diffusion_form, diffusion_norm, last_step, last_step_size, num_steps)
    ("Linear", "velocity", "Euler", "SBDM", 2e-3, "Mean", 0.04, 12),
    ("Linear", "velocity", "Heun", "sigma", 0.7, "Tweedie", 0.04, 9),
    ("Linear", "velocity", "Heun", "SBDM", 1e-3, "Euler", 0.04, 10),
    ("Linear", "velocity", "Euler", "linear", 1.0, None, 0.04, 8),
    ("Linear", "noise", "Euler", "decreasing", 0.05, "Mean", 0.02, 10),
    ("Linear", "score", "Heun", "inccreasing-decreasing", 0.05, "Mean", 0.04, 8),
    ("GVP", "velocity", "Euler", "SBDM", 2e-3, "Mean", 0.04, 10),
    ("VP", "velocity", "Euler", "sigma", 1.0, "Mean", 0.04, 10),
]


def main():
    sys.modules.setdefault("torchdiffeq", types.SimpleNamespace(odeint=None))
    sys.path.insert(0, REF)
    tr = importlib.import_module("transport")
    arrs = {}
    for i, (path, pred, smp, form, norm, last, lss, n) in enumerate(CASES):
        # Transport built directly with eps = 1e-3: create_transport overrides any eps for Linear/GVP + velocity with 0, and
        # the SBDM diffusion is 1/t-singular at t0 = 0 (the reference returns NaN there; nothing to compare)
        mt = {"velocity": tr.ModelType.VELOCITY, "noise": tr.ModelType.NOISE, "score": tr.ModelType.SCORE}[pred]
        pt = {"Linear": tr.PathType.LINEAR, "GVP": tr.PathType.GVP, "VP": tr.PathType.VP}[path]
        t = tr.Transport(model_type=mt, path_type=pt, loss_type=tr.WeightType.NONE, train_eps=1e-3, sample_eps=1e-3)
        fn = tr.Sampler(t).sample_sde(sampling_method=smp, diffusion_form=form, diffusion_norm=norm, last_step=last,
                                      last_step_size=lss, num_steps=n)
        torch.manual_seed(100 + i)
        x0 = torch.randn(3, 2, 4, 4)
        xs = fn(x0, toy_model)
        arrs[f"x0_{i}"] = x0.numpy()
        arrs[f"last_{i}"] = xs[-1].numpy()
        arrs[f"mid_{i}"] = xs[len(xs) // 2].numpy()
        arrs[f"eps_{i}"] = np.array([t.train_eps, t.sample_eps], dtype=np.float64)
        print(i, path, pred, smp, form, last, len(xs), float(xs[-1].abs().mean()))
    arrs["cases"] = np.array(repr(CASES))
    np.savez_compressed(os.path.join(OUT, "sde_sampler.npz"), **arrs)


if __name__ == "__main__":
    main()
