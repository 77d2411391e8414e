"""ODE / SDE integrators of the transport sampler (API mirror of ``lumina_next_t2i/transport/integrators.py``).

The reference delegates ODE stepping to ``torchdiffeq.odeint`` (integrators.py:115; third party, unpinned,
not vendored).  Here:

* when the model callable is a bound ``forward_with_cfg`` / ``forward`` of our engine-backed ``NextDiT`` and
  the drift is the plain velocity field, the whole trajectory runs inside the C-ABI engine
  (``lt_sample_ode``: one host loop in C++, no Python per step, no device syncs);
* for any other callable the fixed-grid solvers of torchdiffeq (euler / midpoint / rk4 "3/8 rule") are
  restated below in plain torch ops on whatever device the state lives on.  This is host glue for the
  model-callable protocol (transport.py:192-195), not a fallback of the HIP path;
* ``dopri5`` (the default ``--solver`` of ``Next-DiT-ImageNet/sample.py:48`` and of ``Sampler.sample_ode``,
  transport.py:349) is torchdiffeq's adaptive Dormand-Prince 5(4) solver, restated below: the step-size controller
  lives on the host (one scalar device->host read per attempted step is inherent to adaptive stepping), every
  model evaluation still runs on the engine through the model callable.  PARITY UNPINNED like the fixed-grid
  solvers (torchdiffeq is neither vendored nor installed); anchored on closed-form ODEs in the tests;
* the other tableau-defined methods ``--solver`` can name (the reference forwards ANY torchdiffeq method string,
  ``lumina_next_t2i/sample.py:77`` -> ``integrators.py:115``): fixed-grid ``heun2`` / ``heun3`` and adaptive ``bosh3`` /
  ``fehlberg2`` / ``adaptive_heun`` run on the same two host loops with their Butcher tableaus (restated from torchdiffeq 0.2.x,
  unpinned like the rest).  ``dopri8``, the Adams multistep family and ``scipy_solver`` are NOT built and say so by name.
"""
import torch as th

FIXED_GRID_METHODS = ("euler", "midpoint", "rk4")           # also built inside the engine (lt_sample_ode): one C-ABI call per trajectory
HOST_FIXED_GRID_METHODS = ("heun2", "heun3")                  # fixed grid, host loop only
ADAPTIVE_METHODS = ("dopri5", "bosh3", "fehlberg2", "adaptive_heun")
NOT_BUILT_METHODS = ("dopri8", "explicit_adams", "implicit_adams", "fixed_adams", "scipy_solver")
ALL_METHODS = FIXED_GRID_METHODS + HOST_FIXED_GRID_METHODS + ADAPTIVE_METHODS


def _unknown_method(method):
    if method in NOT_BUILT_METHODS:
        return NotImplementedError(f"ODE method '{method}' of torchdiffeq is not built here (built: {', '.join(ALL_METHODS)})")
    return ValueError(f"unknown ODE method '{method}' (torchdiffeq names built here: {', '.join(ALL_METHODS)})")


def _call(func, t, y):
    # torchdiffeq's _PerturbFunc casts the time to the state dtype before calling the user function
    return func(t.to(y.dtype), y)


def fixed_grid_odeint(func, y0, t, method="euler"):
    """Solution of dy/dt = func(t, y) at every point of the grid ``t`` (the grid IS the step sequence).

    Restates torchdiffeq.odeint(..., method in {euler, midpoint, rk4}) with no ``step_size`` option:
    euler    y1 = y0 + dt f(t0, y0)
    midpoint y1 = y0 + dt f(t0 + dt/2, y0 + dt/2 f(t0, y0))
    rk4      3/8-rule variant (rk4_alt_step_func): k2 at t0+dt/3, k3 at t0+2dt/3, k4 at t1, weights 1/8 (1,3,3,1)
    heun2    y1 = y0 + dt/2 (f(t0, y0) + f(t1, y0 + dt f(t0, y0)))                                  [tableau alpha 1 | 1 | 1/2 1/2]
    heun3    k2 at t0+dt/3 (y0 + dt/3 k1), k3 at t0+2dt/3 (y0 + 2dt/3 k2), y1 = y0 + dt (k1 + 3 k3) / 4  [1/3 2/3 | 1/3; 0 2/3 | 1/4 0 3/4]
    """
    if method not in FIXED_GRID_METHODS + HOST_FIXED_GRID_METHODS:
        raise _unknown_method(method)
    out = th.empty((len(t),) + tuple(y0.shape), dtype=y0.dtype, device=y0.device)
    out[0] = y0
    y = y0
    third, two_thirds = 1.0 / 3.0, 2.0 / 3.0
    for j in range(len(t) - 1):
        t0, t1 = t[j], t[j + 1]
        dt = t1 - t0
        k1 = _call(func, t0, y)
        if method == "euler":
            dy = dt * k1
        elif method == "midpoint":
            half = 0.5 * dt
            dy = dt * _call(func, t0 + half, y + k1 * half)
        elif method == "heun2":
            k2 = _call(func, t1, y + dt * k1)
            dy = dt * (k1 * 0.5 + k2 * 0.5)
        elif method == "heun3":
            k2 = _call(func, t0 + dt * third, y + dt * k1 * third)
            k3 = _call(func, t0 + dt * two_thirds, y + dt * k2 * two_thirds)
            dy = dt * (k1 * 0.25 + k3 * 0.75)
        else:
            k2 = _call(func, t0 + dt * third, y + dt * k1 * third)
            k3 = _call(func, t0 + dt * two_thirds, y + dt * (k2 - k1 * third))
            k4 = _call(func, t1, y + dt * (k1 - k2 + k3))
            dy = (k1 + 3 * (k2 + k3) + k4) * dt * 0.125
        y = y + dy
        out[j + 1] = y
    return out


# ---- torchdiffeq's adaptive Dormand-Prince 5(4) (rk_common.RKAdaptiveStepsizeODESolver + dopri5.py) -------------------
_DP_ALPHA = (1 / 5, 3 / 10, 4 / 5, 8 / 9, 1.0, 1.0)
_DP_BETA = (
    (1 / 5,),
    (3 / 40, 9 / 40),
    (44 / 45, -56 / 15, 32 / 9),
    (19372 / 6561, -25360 / 2187, 64448 / 6561, -212 / 729),
    (9017 / 3168, -355 / 33, 46732 / 5247, 49 / 176, -5103 / 18656),
    (35 / 384, 0.0, 500 / 1113, 125 / 192, -2187 / 6784, 11 / 84),
)
_DP_C_SOL = (35 / 384, 0.0, 500 / 1113, 125 / 192, -2187 / 6784, 11 / 84, 0.0)
_DP_C_ERR = (35 / 384 - 1951 / 21600, 0.0, 500 / 1113 - 22642 / 50085, 125 / 192 - 451 / 720, -2187 / 6784 + 12231 / 42400,
             11 / 84 - 649 / 6300, -1 / 60)
_DP_C_MID = (6025192743 / 30085553152 / 2, 0.0, 51252292925 / 65400821598 / 2, -2691868925 / 45128329728 / 2,
             187940372067 / 1594534317056 / 2, -1776094331 / 19743644256 / 2, 11237099 / 235043384 / 2)


# (alpha, beta, c_sol, c_err, c_mid, order) of torchdiffeq's other adaptive Runge-Kutta solvers (bosh3.py, fehlberg2.py, adaptive_heun.py)
_TABLEAUS = {
    "dopri5": (_DP_ALPHA, _DP_BETA, _DP_C_SOL, _DP_C_ERR, _DP_C_MID, 5),
    "bosh3": ((1 / 2, 3 / 4, 1.0), ((1 / 2,), (0.0, 3 / 4), (2 / 9, 1 / 3, 4 / 9)), (2 / 9, 1 / 3, 4 / 9, 0.0),
              (2 / 9 - 7 / 24, 1 / 3 - 1 / 4, 4 / 9 - 1 / 3, -1 / 8), (0.0, 0.5, 0.0, 0.0), 3),
    "fehlberg2": ((1 / 2, 1.0), ((1 / 2,), (1 / 256, 255 / 256)), (1 / 512, 255 / 256, 1 / 512), (-1 / 512, 0.0, 1 / 512), (0.0, 0.5, 0.0), 2),
    "adaptive_heun": ((1.0,), ((1.0,),), (0.5, 0.5), (0.5, -0.5), (0.5, 0.0), 2),
}


def _rms(x):
    return x.float().pow(2).mean().sqrt()


def dopri5_odeint(func, y0, t, **kw):
    """torchdiffeq.odeint(..., method="dopri5") - see adaptive_odeint"""
    return adaptive_odeint(func, y0, t, method="dopri5", **kw)


def adaptive_odeint(func, y0, t, *, method="dopri5", rtol=1e-3, atol=1e-6, max_num_steps=2 ** 31 - 1, stats=None, norm=None):
    """torchdiffeq.odeint(func, y0, t, rtol=rtol, atol=atol, method=...) for the adaptive Runge-Kutta family (rk_common.
    RKAdaptiveStepsizeODESolver): solution at every point of ``t`` (dense output through the quartic fitted to y0, y1, a mid-point
    value and the end slopes), steps chosen by the embedded error estimate.  ``method``: dopri5 (5(4), FSAL), bosh3 (3(2), FSAL),
    fehlberg2 (2(1)), adaptive_heun (2(1)); like torchdiffeq, the last stage's slope stands in for f(t1, y1) also where the tableau is
    not first-same-as-last.

    Follows torchdiffeq's controller: initial step from Hairer's heuristic (``_select_initial_step``), error ratio =
    rms(err / (atol + rtol max(|y0|, |y1|))), accept if <= 1, next step = dt * min(10, max(0.9 ratio^-1/5, 0.2 or 1)).
    ``stats`` (optional dict) receives the number of function evaluations and accepted / rejected steps; ``norm`` replaces the
    rms norm of the controller (tuple states use torchdiffeq's mixed norm, see ``tuple_odeint``)."""
    if method not in _TABLEAUS:
        raise _unknown_method(method)
    ALPHA, BETA, C_SOL, C_ERR, C_MID, order = _TABLEAUS[method]
    fsal = C_SOL[-1] == 0.0 and tuple(C_SOL[:-1]) == tuple(BETA[-1])
    nrm = norm if norm is not None else _rms
    t = t.to(device=y0.device)
    tdt = t.dtype

    def f(tt, y):
        return _call(func, tt.to(tdt) if th.is_tensor(tt) else th.as_tensor(tt, dtype=tdt, device=y0.device), y)

    nfe = 0
    t0 = t[0]
    f0 = f(t0, y0)
    nfe += 1
    # _select_initial_step(func, t0, y0, order - 1, ...)
    scale = atol + y0.abs() * rtol
    d0, d1 = nrm(y0 / scale), nrm(f0 / scale)
    h0 = 0.01 * d0 / d1 if (float(d0) >= 1e-5 and float(d1) >= 1e-5) else th.tensor(1e-6, device=y0.device)
    h0 = h0.to(tdt)
    f1 = f(t0 + h0, y0 + h0.to(y0.dtype) * f0)
    nfe += 1
    d2 = nrm((f1 - f0) / scale) / h0
    if float(d1) <= 1e-15 and float(d2) <= 1e-15:
        h1 = th.max(th.tensor(1e-6, dtype=tdt, device=y0.device), h0 * 1e-3)
    else:
        h1 = (0.01 / th.max(d1, d2)) ** (1.0 / float(order))
    dt = th.min(100 * h0, h1.to(tdt)).to(tdt)

    out = th.empty((len(t),) + tuple(y0.shape), dtype=y0.dtype, device=y0.device)
    out[0] = y0
    y, fy, tcur = y0, f0, t0
    tprev = t0
    coeffs = [y0] * 5
    accepted = rejected = 0
    for i in range(1, len(t)):
        next_t = t[i]
        steps = 0
        while float(next_t) > float(tcur):
            assert steps < max_num_steps, "max_num_steps exceeded"
            steps += 1
            t1 = tcur + dt
            dty = dt.to(y.dtype)
            k = [fy]
            yi = y
            for alpha, beta in zip(ALPHA, BETA):
                ti = t1 if alpha == 1.0 else tcur + alpha * dt
                acc = k[0] * beta[0]
                for kj, bj in zip(k[1:], beta[1:]):
                    if bj != 0.0:
                        acc = acc + kj * bj
                yi = y + dty * acc
                k.append(f(ti, yi))
                nfe += 1
            f_1 = k[-1]
            if fsal:  # the last stage IS the solution (dopri5, bosh3)
                y1 = yi
            else:
                sol = k[0] * C_SOL[0]
                for kj, cj in zip(k[1:], C_SOL[1:]):
                    if cj != 0.0:
                        sol = sol + kj * cj
                y1 = y + dty * sol
            err = k[0] * C_ERR[0]
            for kj, cj in zip(k[1:], C_ERR[1:]):
                if cj != 0.0:
                    err = err + kj * cj
            err = dty * err
            tol = atol + rtol * th.max(y.abs(), y1.abs())
            ratio = float(nrm(err / tol))
            if ratio <= 1.0:  # accept: dense-output coefficients of this step, then move on
                mid = k[0] * C_MID[0]
                for kj, cj in zip(k[1:], C_MID[1:]):
                    if cj != 0.0:
                        mid = mid + kj * cj
                y_mid = y + dty * mid
                a = 2 * dty * (f_1 - fy) - 8 * (y1 + y) + 16 * y_mid
                b = dty * (5 * fy - 3 * f_1) + 18 * y + 14 * y1 - 32 * y_mid
                c = dty * (f_1 - 4 * fy) - 11 * y - 5 * y1 + 16 * y_mid
                coeffs = [y, dty * fy, c, b, a]
                tprev, tcur, y, fy = tcur, t1, y1, f_1
                accepted += 1
            else:
                rejected += 1
            # _optimal_step_size(dt, ratio, safety 0.9, ifactor 10, dfactor 0.2, order)
            if ratio == 0.0:
                dt = dt * 10.0
            else:
                dfactor = 1.0 if ratio < 1.0 else 0.2
                dt = dt * min(10.0, max(0.9 / ratio ** (1.0 / order), dfactor))
        # _interp_evaluate(coeffs, tprev, tcur, next_t)
        x = ((next_t - tprev) / (tcur - tprev)).to(y.dtype)
        total = coeffs[0] + x * coeffs[1]
        xp = x
        for cf in coeffs[2:]:
            xp = xp * x
            total = total + xp * cf
        out[i] = total
    if stats is not None:
        stats.update(nfe=nfe, accepted=accepted, rejected=rejected)
    return out


def tuple_odeint(func, y0, t, *, method, rtol=1e-3, atol=1e-6):
    """torchdiffeq.odeint on a TUPLE state (the likelihood ODE's (x, delta_logp), reference transport.py:431-433): the tuple is
    flattened into one 1-d state (``_check_inputs``: cat of reshape(-1)), the user function wrapped (``_TupleFunc``), the solution
    split back into a tuple of [len(t), *shape] tensors; the adaptive controller then measures with the MIXED norm = max over the
    components of their rms norms (``_mixed_norm``).  One tolerance for all components (the reference passes [atol] * len(x))."""
    shapes = [tuple(y.shape) for y in y0]
    sizes = [y.numel() for y in y0]

    def split(flat, lead=()):
        out, o = [], 0
        for sh, n in zip(shapes, sizes):
            out.append(flat[..., o:o + n].reshape(lead + sh))
            o += n
        return tuple(out)

    def f(tt, flat):
        return th.cat([g.reshape(-1) for g in func(tt, split(flat))])

    def mixed_norm(flat):
        return th.stack([_rms(c) for c in split(flat)]).max()

    flat0 = th.cat([y.reshape(-1) for y in y0])
    if method in ADAPTIVE_METHODS:
        sol = adaptive_odeint(f, flat0, t, method=method, rtol=rtol, atol=atol, norm=mixed_norm)
    else:
        sol = fixed_grid_odeint(f, flat0, t, method=method)
    return split(sol, (len(t),))


def _engine_target(model):
    """(NextDiT instance, use_cfg) when ``model`` is one of our engine-backed bound methods, else None."""
    owner = getattr(model, "__self__", None)
    name = getattr(model, "__name__", "")
    if owner is None or not hasattr(owner, "_engine_sample_ode"):
        return None
    if name == "forward_with_cfg":
        return owner, True
    if name == "forward":
        return owner, False
    return None


class ode:
    """ODE solver front end (reference integrators.py:79-116)."""

    def __init__(self, drift, *, t0, t1, sampler_type, num_steps, atol, rtol, time_shifting_factor=None):
        assert t0 < t1, "ODE sampler has to be in forward time"
        self.drift = drift
        self.t = th.linspace(t0, t1, num_steps)
        if time_shifting_factor:
            s = time_shifting_factor
            self.t = self.t / (self.t + s - s * self.t)
        self.atol, self.rtol = atol, rtol
        self.sampler_type = sampler_type
        # torchdiffeq >= 0.2 casts t to the state dtype inside its function wrapper; keep switchable
        self.t_round_to_state_dtype = True

    def sample(self, x, model, **model_kwargs):
        if isinstance(x, tuple):  # the likelihood ODE (reference integrators.py:105-115 with a tuple state)
            device = x[0].device

            def _fn_tuple(t, y):
                return self.drift(y, th.ones(y[0].size(0)).to(device) * t, model, **model_kwargs)

            return tuple_odeint(_fn_tuple, x, self.t.to(device), method=self.sampler_type, rtol=self.rtol, atol=self.atol)
        target = _engine_target(model)
        if (target is not None and getattr(self.drift, "is_plain_velocity", False) and x.is_cuda
                and self.sampler_type in FIXED_GRID_METHODS):
            return self._sample_on_engine(x, target, model_kwargs)

        device = x.device

        def _fn(t, y):
            tvec = th.ones(y.size(0)).to(device) * t  # fp32 [B] (reference integrators.py:108)
            return self.drift(y, tvec, model, **model_kwargs)

        if self.sampler_type in ADAPTIVE_METHODS:
            return adaptive_odeint(_fn, x, self.t.to(device), method=self.sampler_type, rtol=self.rtol, atol=self.atol)
        return fixed_grid_odeint(_fn, x, self.t.to(device), method=self.sampler_type)

    def _sample_on_engine(self, x, target, kw):
        """whole trajectory in one C-ABI call (lt_sample_ode): the model class maps its own forward / forward_with_cfg
        kwargs onto the engine (each reference sub-project has a different kwarg set)"""
        owner, use_cfg = target
        return owner._engine_sample_ode(x, self.t, self.sampler_type, use_cfg, self.t_round_to_state_dtype, dict(kw))


class sde:
    """Euler-Maruyama / Heun SDE sampler (reference integrators.py:5-76); runs through the model callable."""

    def __init__(self, drift, diffusion, *, t0, t1, num_steps, sampler_type):
        assert t0 < t1, "SDE sampler has to be in forward time"
        self.num_timesteps = num_steps
        self.t = th.linspace(t0, t1, num_steps)
        self.dt = self.t[1] - self.t[0]
        self.drift, self.diffusion = drift, diffusion
        if sampler_type not in ("Euler", "Heun"):
            raise NotImplementedError("Smapler type not implemented.")
        self.sampler_type = sampler_type

    def _euler_maruyama(self, x, mean_x, t, model, **kw):
        noise = th.randn(x.size()).to(x)
        tvec = th.ones(x.size(0)).to(x) * t
        dw = noise * th.sqrt(self.dt)
        mean_x = x + self.drift(x, tvec, model, **kw) * self.dt
        return mean_x + th.sqrt(2 * self.diffusion(x, tvec)) * dw, mean_x

    def _heun(self, x, _, t, model, **kw):
        noise = th.randn(x.size()).to(x)
        dw = noise * th.sqrt(self.dt)
        tvec = th.ones(x.size(0)).to(x) * t
        xhat = x + th.sqrt(2 * self.diffusion(x, tvec)) * dw
        k1 = self.drift(xhat, tvec, model, **kw)
        k2 = self.drift(xhat + self.dt * k1, tvec + self.dt, model, **kw)
        return xhat + 0.5 * self.dt * (k1 + k2), xhat

    def sample(self, init, model, **model_kwargs):
        step = self._euler_maruyama if self.sampler_type == "Euler" else self._heun
        x = mean_x = init
        samples = []
        with th.no_grad():
            for ti in self.t[:-1]:
                x, mean_x = step(x, mean_x, ti, model, **model_kwargs)
                samples.append(x)
        return samples
