"""ODE integration for the flow-matching sampler.

Mirrors the reference's `transport/integrators.py` class `ode` (:83-123).  The reference hands the loop to
the third-party `torchdiffeq.odeint` (not vendored, version unpinned, absent from this image), so the solver
is restated here from its published algorithms — see `odeint` below.  Sampler-trajectory parity with
torchdiffeq is therefore UNPINNED; the fixed-grid methods are checked against the CPU oracle and analytic
solutions, the adaptive ones against analytic solutions, the fixed-grid limit and an independent float64 numpy
restatement of the same controller (oracle/zigma_oracle.py, tests/test_host_cpu.py).

Solver state and arithmetic stay in the dtype of `x` (keep the latents fp32; the model casts at its own
boundary), time points in float32 like the reference (`th.linspace`).
"""
import math

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
    # dense output: y(t0 + h/2) = y0 + h * sum(mid_i k_i)  (Shampine's midpoint weights, torchdiffeq's DPS_C_MID)
    mid=[6025192743 / 30085553152 / 2, 0.0, 51252292925 / 65400821598 / 2, -2691868925 / 45128329728 / 2,
         187940372067 / 1594534317056 / 2, -1776094331 / 19743644256 / 2, 11237099 / 235043384 / 2],
    order=5, fsal=True)
_BOSH3 = dict(
    c=[0.0, 1 / 2, 3 / 4, 1.0],
    a=[[], [1 / 2], [0.0, 3 / 4], [2 / 9, 1 / 3, 4 / 9]],
    b=[2 / 9, 1 / 3, 4 / 9, 0.0],
    e=[2 / 9 - 7 / 24, 1 / 3 - 1 / 4, 4 / 9 - 1 / 3, -1 / 8],
    mid=None,
    order=3, fsal=True)
_ADAPTIVE_HEUN = dict(c=[0.0, 1.0], a=[[], [1.0]], b=[0.5, 0.5], e=[0.5, -0.5], mid=None, order=2, fsal=False)
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


def rms_norm(v):
    """torchdiffeq's default norm: sqrt(mean(|v|^2)) over ALL elements (batch included), a 0-dim device tensor."""
    return v.float().pow(2).mean().sqrt()


class AdaptiveStats:
    """Counters of the last adaptive solve (read by tests / tools; not part of the reference API)."""
    nfe = 0
    steps = 0
    accepted = 0
    host_reads = 0


def _adaptive(tab, f, y0, ts, rtol, atol, norm=rms_norm, max_steps=100000):
    """Embedded Runge-Kutta pair with torchdiffeq's controller (RKAdaptiveStepsizeODESolver; restated from the
    published algorithm, the package is absent here):

      * initial step by `_select_initial_step` with exponent 1 / order (torchdiffeq passes order - 1 and uses
        1 / (order' + 1));
      * error ratio = norm(err / (atol + rtol * max(|y0|, |y1|))), accepted iff <= 1;
      * next step = dt * min(10, max(0.9 / ratio^(1/order), dfactor)) with dfactor = 0.2, but 1 when the step was
        accepted with ratio < 1 (an accepted step never shrinks), and dt * 10 when ratio == 0;
      * steps are NOT clamped to the output times or to the end point: the solver steps until t1 >= the requested
        time and evaluates the 4th-order dense output fitted through (y0, y_mid, y1, f0, f1) there (dopri5: Shampine's
        mid-point weights; bosh3 / adaptive_heun: y_mid from the cubic Hermite interpolant, so the fit IS that cubic —
        torchdiffeq's own mid-point weights for these two low-order pairs could not be checked here).

    Step control runs ON THE DEVICE: error norm, step factor, accepted/rejected selection of (y, f, t) and the step
    size are 0-dim device tensors; the host reads ONE small tensor (accepted?, t1) per step, which it needs to know
    which output times the step covered and when to stop.  Time is float64 like torchdiffeq's; the model sees float32."""
    order, a, b, c, e, mid = tab["order"], tab["a"], tab["b"], tab["c"], tab["e"], tab["mid"]
    dev = y0.device
    f64 = dict(device=dev, dtype=th.float64)
    ts_host = [float(v) for v in ts.tolist()]                        # output times are host data to begin with
    if len(ts_host) > 1 and ts_host[0] > ts_host[-1]:
        # decreasing time grid (Sampler.sample_ode(reverse=True): data -> noise): integrate y'(s) = -f(-s, y) on s = -t, as
        # torchdiffeq does (odeint's `_check_inputs` flips the sign of t and of the function)
        return _adaptive(tab, lambda s, y: -f(-s, y), y0, -ts, rtol, atol, norm=norm, max_steps=max_steps)
    if any(t1 <= t0 for t0, t1 in zip(ts_host, ts_host[1:])):
        raise ValueError("t must be strictly increasing or strictly decreasing")
    call = lambda t, y: f(t.to(th.float32), y)
    t0 = th.tensor(ts_host[0], **f64)
    f0 = call(t0, y0)
    # -- _select_initial_step, all on the device --------------------------------------------------------
    scale = atol + y0.abs() * rtol
    d0, d1 = norm(y0 / scale).double(), norm(f0 / scale).double()
    h0 = th.where((d0 < 1e-5) | (d1 < 1e-5), th.full((), 1e-6, **f64), 0.01 * d0 / d1.clamp_min(1e-300))
    f1 = call(t0 + h0, y0 + h0 * f0)
    d2 = norm((f1 - f0) / scale).double() / h0
    dmax = th.maximum(d1, d2)
    h1 = th.where(dmax <= 1e-15, th.clamp_min(h0 * 1e-3, 1e-6), (0.01 / dmax.clamp_min(1e-300)) ** (1.0 / order))
    dt = th.minimum(100 * h0, h1)
    st = AdaptiveStats
    st.nfe, st.steps, st.accepted, st.host_reads = 2, 0, 0, 0
    out = [y0]
    nxt = 1
    t0_host = ts_host[0]
    y = y0
    ten, one, fifth = th.full((), 10.0, **f64), th.full((), 1.0, **f64), th.full((), 0.2, **f64)
    while nxt < len(ts_host):
        if st.steps >= max_steps:
            raise RuntimeError("adaptive ODE solver exceeded max_steps")
        st.steps += 1
        ks = [f0]
        for i in range(1, len(c)):
            yi = y
            for j, aij in enumerate(a[i]):
                if aij != 0.0:
                    yi = yi + (dt * aij) * ks[j]
            ks.append(call(t0 + c[i] * dt, yi))
        st.nfe += len(c) - 1
        y1, err, y_mid = y, None, y
        for bi, ei, mi, k in zip(b, e, mid or [0.0] * len(b), ks):
            if bi != 0.0:
                y1 = y1 + (dt * bi) * k
            if ei != 0.0:
                err = (dt * ei) * k if err is None else err + (dt * ei) * k
            if mi != 0.0:
                y_mid = y_mid + (dt * mi) * k
        ratio = norm(err / (atol + rtol * th.maximum(y.abs(), y1.abs()))).double()
        accept = ratio <= 1
        dfac = th.where(ratio < 1, one, fifth)
        factor = th.where(ratio == 0, ten, th.minimum(ten, th.maximum(0.9 / ratio.clamp_min(1e-300) ** (1.0 / order), dfac)))
        t1 = t0 + dt
        flag, t1_host, ratio_host = th.stack([accept.double(), t1, ratio]).tolist()      # the ONE host read of this step
        st.host_reads += 1
        # torchdiffeq asserts on a non-finite state / step underflow; without this a NaN from the model (accept is then always
        # False and dt becomes NaN) would spin to max_steps at 6 model evaluations + one host read per step
        if ratio_host != ratio_host or not math.isfinite(t1_host):
            raise RuntimeError("adaptive ODE solver: non-finite error estimate or time (the model returned NaN / inf?)")
        if not t1_host > t0_host:
            raise RuntimeError("adaptive ODE solver: step size underflow")
        if flag:
            st.accepted += 1
            f_new = ks[-1] if tab["fsal"] else call(t1, y1)
            st.nfe += 0 if tab["fsal"] else 1
            if nxt < len(ts_host) and ts_host[nxt] <= t1_host:
                # dense output (torchdiffeq _interp_fit / _interp_evaluate): quartic through y0, y_mid, y1, f0, f1
                if mid is None:       # no dense-output weights: the value the cubic Hermite interpolant takes at t0 + dt/2
                    y_mid = 0.5 * (y + y1) + (dt * 0.125) * (f0 - f_new)
                ca = 2 * dt * (f_new - f0) - 8 * (y1 + y) + 16 * y_mid
                cb = dt * (5 * f0 - 3 * f_new) + 18 * y + 14 * y1 - 32 * y_mid
                cc = dt * (f_new - 4 * f0) - 11 * y - 5 * y1 + 16 * y_mid
                cd = dt * f0
                while nxt < len(ts_host) and ts_host[nxt] <= t1_host:
                    s = (ts_host[nxt] - t0_host) / (t1_host - t0_host)
                    out.append((y + s * cd + s ** 2 * cc + s ** 3 * cb + s ** 4 * ca).to(y0.dtype))
                    nxt += 1
            y, f0, t0, t0_host = y1, f_new, t1, t1_host
        dt = dt * factor
    return th.stack(out), st.nfe


def odeint(func, y0, t, *, method="dopri5", rtol=1e-3, atol=1e-6, norm=rms_norm):
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
        return _adaptive(_ADAPTIVE[method], func, y0, t, rtol, atol, norm=norm)[0]
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

        def mixed_norm(v):        # torchdiffeq's default for tuple states: max over the components' RMS norms
            parts = th.split(v, sizes, dim=1)
            return th.stack([rms_norm(q) for q in parts]).max()

        ys = odeint(_fn_tuple, pack(x), t, method=self.sampler_type, atol=[self.atol], rtol=[self.rtol], norm=mixed_norm)
        return tuple(q.reshape((ys.shape[0],) + tuple(sh)) for q, sh in zip(th.split(ys, sizes, dim=2), shapes))
