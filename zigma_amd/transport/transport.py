"""Flow-matching transport: training target, drift / score conversion, samplers.

Mirrors the reference's `transport/transport.py`: `ModelType/PathType/WeightType` (:18-40),
`Transport` (:43-233) and `Sampler` (:236-478).  `Sampler.sample_ode` is the hot path (called by
sample_acc.py:159-165,362); the likelihood sampler is a later scope row and raises.
"""
import enum

import numpy as np
import torch as th

from . import path
from .integrators import ode, sde
from .path import expand_t_like_x


class ModelType(enum.Enum):
    NOISE = enum.auto()
    SCORE = enum.auto()
    VELOCITY = enum.auto()


class PathType(enum.Enum):
    LINEAR = enum.auto()
    GVP = enum.auto()
    VP = enum.auto()


class WeightType(enum.Enum):
    NONE = enum.auto()
    VELOCITY = enum.auto()
    LIKELIHOOD = enum.auto()


def mean_flat(x):
    return th.mean(x, dim=list(range(1, len(x.size()))))


class Transport:
    def __init__(self, *, model_type, path_type, loss_type, train_eps, sample_eps):
        plans = {PathType.LINEAR: path.ICPlan, PathType.GVP: path.GVPCPlan, PathType.VP: path.VPCPlan}
        self.loss_type = loss_type
        self.model_type = model_type
        self.path_sampler = plans[path_type]()
        self.train_eps = train_eps
        self.sample_eps = sample_eps

    def prior_logp(self, z):
        """log-density of the standard normal prior, per sample (reference transport.py:69-77)."""
        n = z[0].numel()
        return -n / 2.0 * np.log(2 * np.pi) - z.reshape(z.shape[0], -1).pow(2).sum(1) / 2.0

    def check_interval(self, train_eps, sample_eps, *, diffusion_form="SBDM", sde=False, reverse=False, eval=False,
                       last_step_size=0.0):
        t0, t1 = 0, 1
        eps = train_eps if not eval else sample_eps
        if type(self.path_sampler) in [path.VPCPlan]:
            t1 = 1 - eps if (not sde or last_step_size == 0) else 1 - last_step_size
        elif type(self.path_sampler) in [path.ICPlan, path.GVPCPlan] and (self.model_type != ModelType.VELOCITY or sde):
            t0 = eps if (diffusion_form == "SBDM" and sde) or self.model_type != ModelType.VELOCITY else 0
            t1 = 1 - eps if (not sde or last_step_size == 0) else 1 - last_step_size
        if reverse:
            t0, t1 = 1 - t0, 1 - t1
        return t0, t1

    def sample(self, x1):
        """Draw (t, x0) for a data batch x1."""
        x0 = th.randn_like(x1)
        t0, t1 = self.check_interval(self.train_eps, self.sample_eps)
        t = (th.rand((x1.shape[0],)) * (t1 - t0) + t0).to(x1)
        return t, x0, x1

    def training_losses(self, model, x1, model_kwargs=None):
        model_kwargs = model_kwargs or {}
        t, x0, x1 = self.sample(x1)
        t, xt, ut = self.path_sampler.plan(t, x0, x1)
        model_output = model(xt, t, **model_kwargs)
        assert model_output.size() == xt.size(), "Model output shape does not match input shape"
        terms = {"pred": model_output}
        if self.model_type == ModelType.VELOCITY:
            terms["loss"] = mean_flat((model_output - ut) ** 2)
            return terms
        _, drift_var = self.path_sampler.compute_drift(xt, t)
        sigma_t, _ = self.path_sampler.compute_sigma_t(expand_t_like_x(t, xt))
        if self.loss_type == WeightType.VELOCITY:
            weight = (drift_var / sigma_t) ** 2
        elif self.loss_type == WeightType.LIKELIHOOD:
            weight = drift_var / (sigma_t ** 2)
        elif self.loss_type == WeightType.NONE:
            weight = 1
        else:
            raise NotImplementedError()
        if self.model_type == ModelType.NOISE:
            terms["loss"] = mean_flat(weight * ((model_output - x0) ** 2))
        elif self.model_type == ModelType.SCORE:
            terms["loss"] = mean_flat(weight * ((model_output * sigma_t + x0) ** 2))
        else:
            raise NotImplementedError()
        return terms

    def get_drift(self):
        """drift of the probability-flow ODE for the configured parametrisation."""
        ps = self.path_sampler

        def score_ode(x, t, model, **kw):
            drift_mean, drift_var = ps.compute_drift(x, t)
            return -drift_mean + drift_var * model(x, t, **kw)

        def noise_ode(x, t, model, **kw):
            drift_mean, drift_var = ps.compute_drift(x, t)
            sigma_t, _ = ps.compute_sigma_t(expand_t_like_x(t, x))
            return -drift_mean + drift_var * (model(x, t, **kw) / -sigma_t)

        def velocity_ode(x, t, model, **kw):
            return model(x, t, **kw)

        drift_fn = {ModelType.NOISE: noise_ode, ModelType.SCORE: score_ode, ModelType.VELOCITY: velocity_ode}[self.model_type]

        def body_fn(x, t, model, **kw):
            out = drift_fn(x, t, model, **kw)
            assert out.shape == x.shape, "Output shape from ODE solver must match input shape"
            return out

        return body_fn

    def get_score(self):
        ps = self.path_sampler
        if self.model_type == ModelType.NOISE:
            return lambda x, t, model, **kw: model(x, t, **kw) / -ps.compute_sigma_t(expand_t_like_x(t, x))[0]
        if self.model_type == ModelType.SCORE:
            return lambda x, t, model, **kw: model(x, t, **kw)
        if self.model_type == ModelType.VELOCITY:
            return lambda x, t, model, **kw: ps.get_score_from_velocity(model(x, t, **kw), x, t)
        raise NotImplementedError()


class Sampler:
    """Sampler class for the transport model."""

    def __init__(self, transport):
        self.transport = transport
        self.drift = self.transport.get_drift()
        self.score = self.transport.get_score()

    def _sde_drift_and_diffusion(self, *, diffusion_form="SBDM", diffusion_norm=1.0):
        """drift + w_t * score and w_t (reference transport.py:251-271).  The reference evaluates the network twice per
        drift call (once inside self.drift, once inside self.score, same arguments); the network is deterministic, so
        it is evaluated ONCE here and both closures read that output — half the denoiser forwards per SDE step."""
        ps = self.transport.path_sampler

        def diffusion_fn(x, t):
            return ps.compute_diffusion(x, t, form=diffusion_form, norm=diffusion_norm)

        def sde_drift(x, t, model, **kw):
            out = model(x, t, **kw)
            cached = lambda *_a, **_k: out
            return self.drift(x, t, cached) + diffusion_fn(x, t) * self.score(x, t, cached)

        return sde_drift, diffusion_fn

    def _last_step(self, sde_drift, *, last_step, last_step_size):
        """(reference transport.py:273-307)"""
        if last_step is None:
            return lambda x, t, model, **kw: x
        if last_step == "Mean":
            return lambda x, t, model, **kw: x + sde_drift(x, t, model, **kw) * last_step_size
        if last_step == "Tweedie":
            alpha, sigma = self.transport.path_sampler.compute_alpha_t, self.transport.path_sampler.compute_sigma_t
            return lambda x, t, model, **kw: (x / alpha(t)[0][0]
                                              + (sigma(t)[0][0] ** 2) / alpha(t)[0][0] * self.score(x, t, model, **kw))
        if last_step == "Euler":
            return lambda x, t, model, **kw: x + self.drift(x, t, model, **kw) * last_step_size
        raise NotImplementedError()

    def sample_sde(self, *, sampling_method="Euler", diffusion_form="SBDM", diffusion_norm=1.0, last_step="Mean",
                   last_step_size=0.04, num_steps=250):
        """returns fn(init_z, model, **model_kwargs) -> list of num_steps states (reference transport.py:309-370):
        num_steps - 1 solver steps on linspace(t0, t1, num_steps) plus the last step ("Mean" | "Tweedie" | "Euler" | None)
        from t1 = 1 - last_step_size."""
        if last_step is None:
            last_step_size = 0.0
        sde_drift, sde_diffusion = self._sde_drift_and_diffusion(diffusion_form=diffusion_form,
                                                                  diffusion_norm=diffusion_norm)
        t0, t1 = self.transport.check_interval(self.transport.train_eps, self.transport.sample_eps,
                                               diffusion_form=diffusion_form, sde=True, eval=True, reverse=False,
                                               last_step_size=last_step_size)
        _sde = sde(sde_drift, sde_diffusion, t0=t0, t1=t1, num_steps=num_steps, sampler_type=sampling_method)
        last_step_fn = self._last_step(sde_drift, last_step=last_step, last_step_size=last_step_size)

        def _sample(init_z, model, **model_kwargs):
            xs = _sde.sample(init_z, model, **model_kwargs)
            ts = th.ones(init_z.size(0), device=init_z.device) * t1
            with th.no_grad():
                xs.append(last_step_fn(xs[-1], ts, model, **model_kwargs))
            assert len(xs) == num_steps, "Samples does not match the number of steps"
            return xs

        return _sample

    def sample_ode(self, *, sampling_method="dopri5", num_steps=50, atol=1e-6, rtol=1e-3, reverse=False):
        """returns fn(x, model, **model_kwargs) -> Tensor(num_steps, *x.shape); callers take [-1].
        Fixed-grid methods (euler, midpoint, heun2, heun3, rk4) do num_steps-1 steps on linspace(t0, t1, num_steps);
        adaptive ones (dopri5, bosh3, adaptive_heun) report at those times."""
        if reverse:
            drift = lambda x, t, model, **kw: self.drift(x, th.ones_like(t) * (1 - t), model, **kw)
        else:
            drift = self.drift
        t0, t1 = self.transport.check_interval(self.transport.train_eps, self.transport.sample_eps, sde=False,
                                               eval=True, reverse=reverse, last_step_size=0.0)
        return ode(drift=drift, t0=t0, t1=t1, sampler_type=sampling_method, num_steps=num_steps, atol=atol,
                   rtol=rtol).sample

    def sample_ode_likelihood(self, *, sampling_method="dopri5", num_steps=50, atol=1e-6, rtol=1e-3):
        """returns fn(x, model, **model_kwargs) -> (logp, z): the data-to-noise ODE integrated together with the
        Hutchinson estimate of the divergence (reference transport.py:419-478).  Needs a vjp through the model: with
        `zigma_amd.model_zigma.ZigMa` that runs through the HIP backward kernels.  The reference evaluates the drift
        twice per call (once under autograd, once for the value); one evaluation serves both here."""

        def _likelihood_drift(x, t, model, **model_kwargs):
            x, _ = x
            eps = th.randint(2, x.size(), dtype=th.float, device=x.device).to(x.dtype) * 2 - 1
            t = th.ones_like(t) * (1 - t)
            with th.enable_grad():
                x = x.detach().requires_grad_(True)
                drift = self.drift(x, t, model, **model_kwargs)
                grad = th.autograd.grad(th.sum(drift * eps), x)[0]
            logp_grad = th.sum(grad * eps, dim=tuple(range(1, len(x.size()))))
            return (-drift.detach(), logp_grad)

        t0, t1 = self.transport.check_interval(self.transport.train_eps, self.transport.sample_eps, sde=False, eval=True,
                                               reverse=False, last_step_size=0.0)
        _ode = ode(drift=_likelihood_drift, t0=t0, t1=t1, sampler_type=sampling_method, num_steps=num_steps, atol=atol,
                   rtol=rtol)

        def _sample_fn(x, model, **model_kwargs):
            init_logp = th.zeros(x.size(0)).to(x)
            drift, delta_logp = _ode.sample((x, init_logp), model, **model_kwargs)
            drift, delta_logp = drift[-1], delta_logp[-1]
            prior_logp = self.transport.prior_logp(drift)
            return prior_logp - delta_logp, drift

        return _sample_fn
