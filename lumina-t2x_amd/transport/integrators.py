"""ODE / SDE integrators of the transport sampler (API mirror of ``lumina_next_t2i/transport/integrators.py``).

The reference delegates ODE stepping to ``torchdiffeq.odeint`` (integrators.py:115; third party, unpinned,
not vendored).  Here:

* when the model callable is a bound ``forward_with_cfg`` / ``forward`` of our engine-backed ``NextDiT`` and
  the drift is the plain velocity field, the whole trajectory runs inside the C-ABI engine
  (``lt_sample_ode``: one host loop in C++, no Python per step, no device syncs);
* for any other callable the fixed-grid solvers of torchdiffeq (euler / midpoint / rk4 "3/8 rule") are
  restated below in plain torch ops on whatever device the state lives on.  This is host glue for the
  model-callable protocol (transport.py:192-195), not a fallback of the HIP path.
"""
import torch as th

FIXED_GRID_METHODS = ("euler", "midpoint", "rk4")


def _call(func, t, y):
    # torchdiffeq's _PerturbFunc casts the time to the state dtype before calling the user function
    return func(t.to(y.dtype), y)


def fixed_grid_odeint(func, y0, t, method="euler"):
    """Solution of dy/dt = func(t, y) at every point of the grid ``t`` (the grid IS the step sequence).

    Restates torchdiffeq.odeint(..., method in {euler, midpoint, rk4}) with no ``step_size`` option:
    euler    y1 = y0 + dt f(t0, y0)
    midpoint y1 = y0 + dt f(t0 + dt/2, y0 + dt/2 f(t0, y0))
    rk4      3/8-rule variant (rk4_alt_step_func): k2 at t0+dt/3, k3 at t0+2dt/3, k4 at t1, weights 1/8 (1,3,3,1)
    """
    if method not in FIXED_GRID_METHODS:
        raise NotImplementedError(
            f"ODE method '{method}': only the fixed-grid solvers {FIXED_GRID_METHODS} are built "
            "(adaptive dopri5 is listed as a later row in SURVEY.md 8f)")
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
        else:
            k2 = _call(func, t0 + dt * third, y + dt * k1 * third)
            k3 = _call(func, t0 + dt * two_thirds, y + dt * (k2 - k1 * third))
            k4 = _call(func, t1, y + dt * (k1 - k2 + k3))
            dy = (k1 + 3 * (k2 + k3) + k4) * dt * 0.125
        y = y + dy
        out[j + 1] = y
    return out


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
        if isinstance(x, tuple):
            raise NotImplementedError("tuple states (likelihood ODE) are a later row (SURVEY.md 8f)")
        target = _engine_target(model)
        if (target is not None and getattr(self.drift, "is_plain_velocity", False) and x.is_cuda
                and self.sampler_type in FIXED_GRID_METHODS):
            return self._sample_on_engine(x, target, model_kwargs)

        device = x.device

        def _fn(t, y):
            tvec = th.ones(y.size(0)).to(device) * t  # fp32 [B] (reference integrators.py:108)
            return self.drift(y, tvec, model, **model_kwargs)

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
