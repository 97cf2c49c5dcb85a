"""ODE integration for the flow-matching sampler.

Mirrors the reference's `transport/integrators.py` class `ode` (:83-123).  The reference hands the loop to
the third-party `torchdiffeq.odeint` (not vendored, version unpinned, absent from this image), so the solver
is restated here from its published algorithms — see `odeint` below.  Sampler-trajectory parity with
torchdiffeq is therefore UNPINNED; the fixed-grid methods are checked against the CPU oracle and analytic
solutions, dopri5 against analytic solutions and the fixed-grid limit.

Solver state and arithmetic stay in the dtype of `x` (keep the latents fp32; the model casts at its own
boundary), time points in float32 like the reference (`th.linspace`).
"""
import torch as th

# -- Butcher tableaux ---------------------------------------------------------------------------------
_DOPRI5 = dict(
    c=[0.0, 1 / 5, 3 / 10, 4 / 5, 8 / 9, 1.0, 1.0],
    a=[[], [1 / 5], [3 / 40, 9 / 40], [44 / 45, -56 / 15, 32 / 9],
       [19372 / 6561, -25360 / 2187, 64448 / 6561, -212 / 729],
       [9017 / 3168, -355 / 33, 46732 / 5247, 49 / 176, -5103 / 18656],
       [35 / 384, 0.0, 500 / 1113, 125 / 192, -2187 / 6784, 11 / 84]],
    b=[35 / 384, 0.0, 500 / 1113, 125 / 192, -2187 / 6784, 11 / 84, 0.0],
    e=[35 / 384 - 1951 / 21600, 0.0, 500 / 1113 - 22642 / 50085, 125 / 192 - 451 / 720,
       -2187 / 6784 + 12231 / 42400, 11 / 84 - 649 / 6300, -1.0 / 60.0],
    order=5, fsal=True)
_BOSH3 = dict(
    c=[0.0, 1 / 2, 3 / 4, 1.0],
    a=[[], [1 / 2], [0.0, 3 / 4], [2 / 9, 1 / 3, 4 / 9]],
    b=[2 / 9, 1 / 3, 4 / 9, 0.0],
    e=[2 / 9 - 7 / 24, 1 / 3 - 1 / 4, 4 / 9 - 1 / 3, -1 / 8],
    order=3, fsal=True)
_ADAPTIVE_HEUN = dict(c=[0.0, 1.0], a=[[], [1.0]], b=[0.5, 0.5], e=[0.5, -0.5], order=2, fsal=False)
_ADAPTIVE = {"dopri5": _DOPRI5, "bosh3": _BOSH3, "adaptive_heun": _ADAPTIVE_HEUN}


def _fixed_step(method, f, t0, t1, x):
    h = t1 - t0
    if method == "euler":
        return x + h * f(t0, x)
    if method == "midpoint":
        return x + h * f(t0 + 0.5 * h, x + 0.5 * h * f(t0, x))
    if method in ("heun", "heun2"):
        k1 = f(t0, x)
        return x + 0.5 * h * (k1 + f(t1, x + h * k1))
    if method == "heun3":
        k1 = f(t0, x)
        k2 = f(t0 + h / 3, x + h / 3 * k1)
        k3 = f(t0 + 2 * h / 3, x + 2 * h / 3 * k2)
        return x + h * (0.25 * k1 + 0.75 * k3)
    if method == "rk4":          # 3/8 rule, as torchdiffeq's fixed-grid rk4
        k1 = f(t0, x)
        k2 = f(t0 + h / 3, x + h * k1 / 3)
        k3 = f(t0 + h * 2 / 3, x + h * (k2 - k1 / 3))
        k4 = f(t1, x + h * (k1 - k2 + k3))
        return x + h * (k1 + 3 * (k2 + k3) + k4) / 8
    raise ValueError(f"unknown fixed-grid method {method}")


FIXED_GRID = ("euler", "midpoint", "heun", "heun2", "heun3", "rk4")


def _rms(v):
    return v.float().pow(2).mean().sqrt()


def _adaptive(tab, f, x, ts, rtol, atol, max_steps=100000):
    """Embedded Runge-Kutta with the standard step controller (Hairer-Norsett-Wanner II.4): error norm =
    RMS of err / (atol + rtol * max(|y0|, |y1|)); step factor 0.9 * ratio^(-1/order) clamped to [0.2, 10];
    outputs at the requested times by cubic Hermite interpolation inside the accepted step."""
    order, a, b, c, e = tab["order"], tab["a"], tab["b"], tab["c"], tab["e"]
    t = float(ts[0])
    t_end = float(ts[-1])
    direction = 1.0 if t_end >= t else -1.0
    f0 = f(ts[0], x)
    # initial step (Hairer): h0 = 0.01 |y|/|f|, refined by one Euler probe
    scale = atol + rtol * x.abs()
    d0, d1 = float(_rms(x / scale)), float(_rms(f0 / scale))
    h0 = 1e-6 if d0 < 1e-5 or d1 < 1e-5 else 0.01 * d0 / d1
    f1 = f(ts[0] + direction * h0, x + direction * h0 * f0)
    d2 = float(_rms((f1 - f0) / scale)) / h0
    h1 = max(1e-6, h0 * 1e-3) if max(d1, d2) <= 1e-15 else (0.01 / max(d1, d2)) ** (1.0 / (order + 1))
    h = direction * min(100 * h0, h1)
    out = [x]
    nxt = 1
    nfe = 2
    for _ in range(max_steps):
        if nxt >= len(ts):
            break
        if (t + h - t_end) * direction > 0:
            h = t_end - t
        ks = [f0]
        for i in range(1, len(c)):
            xi = x
            for j, aij in enumerate(a[i]):
                if aij != 0.0:
                    xi = xi + (h * aij) * ks[j]
            ks.append(f(x.new_tensor(t + c[i] * h, dtype=th.float32), xi))
        nfe += len(c) - 1
        x1 = x
        err = th.zeros_like(x)
        for bi, ei, k in zip(b, e, ks):
            if bi != 0.0:
                x1 = x1 + (h * bi) * k
            if ei != 0.0:
                err = err + (h * ei) * k
        tol = atol + rtol * th.maximum(x.abs(), x1.abs())
        ratio = float(_rms(err / tol))
        if ratio <= 1.0:
            f_new = ks[-1] if tab["fsal"] else f(x.new_tensor(t + h, dtype=th.float32), x1)
            nfe += 0 if tab["fsal"] else 1
            while nxt < len(ts) and (float(ts[nxt]) - (t + h)) * direction <= 1e-12:
                s = (float(ts[nxt]) - t) / h                # cubic Hermite on [t, t + h]
                h00, h10 = 2 * s ** 3 - 3 * s ** 2 + 1, s ** 3 - 2 * s ** 2 + s
                h01, h11 = -2 * s ** 3 + 3 * s ** 2, s ** 3 - s ** 2
                out.append(x1 if abs(s - 1.0) < 1e-12 else h00 * x + (h10 * h) * f0 + h01 * x1 + (h11 * h) * f_new)
                nxt += 1
            t, x, f0 = t + h, x1, f_new
        factor = 10.0 if ratio == 0.0 else min(10.0, max(0.2, 0.9 * ratio ** (-1.0 / order)))
        h = h * factor
    else:
        raise RuntimeError("adaptive ODE solver exceeded max_steps")
    return th.stack(out), nfe


def odeint(func, y0, t, *, method="dopri5", rtol=1e-3, atol=1e-6):
    """Solve y' = func(t, y), y(t[0]) = y0; returns the solution at every t[i], shape (len(t), *y0.shape).
    Fixed-grid methods step exactly on the grid `t`; adaptive ones only report there."""
    rtol = rtol[0] if isinstance(rtol, (list, tuple)) else rtol
    atol = atol[0] if isinstance(atol, (list, tuple)) else atol
    if method in FIXED_GRID:
        ys = [y0]
        y = y0
        for k in range(len(t) - 1):
            y = _fixed_step(method, func, t[k], t[k + 1], y)
            ys.append(y)
        return th.stack(ys)
    if method in _ADAPTIVE:
        return _adaptive(_ADAPTIVE[method], func, y0, t, rtol, atol)[0]
    raise ValueError(f"sampling_method {method!r} is not implemented (have {FIXED_GRID + tuple(_ADAPTIVE)})")


class sde:
    """SDE solver class (reference transport/integrators.py:9-80): Euler-Maruyama and the stochastic Heun scheme on
    linspace(t0, t1, num_steps).  The Wiener increments are drawn with th.randn on the CPU generator and moved to the
    state's device/dtype, exactly like the reference (`th.randn(x.size()).to(x)`), so a seeded run reproduces the
    reference's noise stream; everything else stays on the device."""

    def __init__(self, drift, diffusion, *, t0, t1, num_steps, sampler_type):
        assert t0 < t1, "SDE sampler has to be in forward time"
        self.num_timesteps = num_steps
        self.t = th.linspace(t0, t1, num_steps)
        self.dt = self.t[1] - self.t[0]
        self.drift = drift
        self.diffusion = diffusion
        self.sampler_type = sampler_type

    @staticmethod
    def _sqrt2(d, x):
        # diffusion_form="constant" hands back a Python float (the reference's th.sqrt raises on it)
        return th.sqrt(2 * (d if th.is_tensor(d) else th.as_tensor(d, dtype=x.dtype, device=x.device)))

    def _euler_maruyama_step(self, x, mean_x, t, model, **model_kwargs):
        w_cur = th.randn(x.size()).to(x)
        dt = self.dt.to(x)
        t = th.ones(x.size(0)).to(x) * t
        dw = w_cur * th.sqrt(dt)
        drift = self.drift(x, t, model, **model_kwargs)
        diffusion = self.diffusion(x, t)
        mean_x = x + drift * dt
        x = mean_x + self._sqrt2(diffusion, x) * dw
        return x, mean_x

    def _heun_step(self, x, _, t, model, **model_kwargs):
        w_cur = th.randn(x.size()).to(x)
        dt = self.dt.to(x)
        dw = w_cur * th.sqrt(dt)
        t_cur = th.ones(x.size(0)).to(x) * t
        diffusion = self.diffusion(x, t_cur)
        xhat = x + self._sqrt2(diffusion, x) * dw
        K1 = self.drift(xhat, t_cur, model, **model_kwargs)
        xp = xhat + dt * K1
        K2 = self.drift(xp, t_cur + dt, model, **model_kwargs)
        return xhat + 0.5 * dt * (K1 + K2), xhat      # at the last time point no Heun step is performed

    def sample(self, init, model, **model_kwargs):
        """forward loop of the SDE -> list of num_steps - 1 states"""
        try:
            sampler = {"Euler": self._euler_maruyama_step, "Heun": self._heun_step}[self.sampler_type]
        except KeyError:
            raise NotImplementedError("Sampler type not implemented.")
        x, mean_x, samples = init, init, []
        for ti in self.t[:-1]:
            with th.no_grad():
                x, mean_x = sampler(x, mean_x, ti, model, **model_kwargs)
                samples.append(x)
        return samples


class ode:
    """ODE solver class (reference transport/integrators.py:83-123)."""

    def __init__(self, drift, *, t0, t1, sampler_type, num_steps, atol, rtol):
        self.drift = drift
        self.t = th.linspace(t0, t1, num_steps)
        self.atol = atol
        self.rtol = rtol
        self.sampler_type = sampler_type

    def sample(self, x, model, **model_kwargs):
        """x: Tensor -> Tensor(num_steps, *x.shape);  tuple of per-sample tensors (the likelihood ODE's (x, logp)) ->
        tuple of such tensors.  A tuple state is integrated as ONE flattened (batch, sum of sizes) vector, like
        torchdiffeq does (one shared step size and error norm)."""
        is_tuple = isinstance(x, tuple)
        device = x[0].device if is_tuple else x.device
        t = self.t.to(device)
        if not is_tuple:
            ones = th.ones(x.size(0), device=device)

            def _fn(t, x):
                return self.drift(x, ones * t, model, **model_kwargs)

            return odeint(_fn, x, t, method=self.sampler_type, atol=[self.atol], rtol=[self.rtol])
        bsz = x[0].size(0)
        shapes = [xi.shape for xi in x]
        sizes = [xi[0].numel() if xi.dim() > 1 else 1 for xi in x]
        ones = th.ones(bsz, device=device)
        pack = lambda parts: th.cat([q.reshape(bsz, -1) for q in parts], dim=1)
        unpack = lambda y: tuple(q.reshape(sh) for q, sh in zip(th.split(y, sizes, dim=1), shapes))

        def _fn_tuple(t, y):
            return pack(self.drift(unpack(y), ones * t, model, **model_kwargs))

        ys = odeint(_fn_tuple, pack(x), t, method=self.sampler_type, atol=[self.atol], rtol=[self.rtol])
        return tuple(q.reshape((ys.shape[0],) + tuple(sh)) for q, sh in zip(th.split(ys, sizes, dim=2), shapes))
