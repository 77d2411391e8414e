"""Fixed-grid ODE solvers restated from torchdiffeq (euler / midpoint / rk4).  TEST INFRASTRUCTURE ONLY.

PARITY UNPINNED: torchdiffeq is a third-party dependency of the reference (call site
``lumina_next_t2i/transport/integrators.py:115``; unpinned in pyproject.toml:50 / requirements.txt:7) that is
neither vendored nor installed here.  Restated from its published algorithm (v0.2.x ``FixedGridODESolver``):
the output grid is the step grid; each user-function call goes through ``_PerturbFunc`` which casts ``t`` to
the state dtype; ``rk4`` is the 3/8-rule ``rk4_alt_step_func``.  Anchors: the reference's own midpoint step
``visual_anagrams/generate.py:212-219`` and closed-form known answers in tests/test_oracle_golden.py.
"""
import torch


def _odeint_tuple(func, y0, t, **kw):
    """tuple states (the likelihood ODE, transport.py:431-433): torchdiffeq flattens the tuple into ONE 1-d state
    (``_check_inputs``: cat of reshape(-1)), wraps the user function (``_TupleFunc``) and splits the solution back."""
    shapes = [tuple(y.shape) for y in y0]
    sizes = [y.numel() for y in y0]

    def split(flat, lead=()):
        out, o = [], 0
        for sh, n in zip(shapes, sizes):
            out.append(flat[..., o:o + n].reshape(lead + sh))
            o += n
        return tuple(out)

    def f(tt, flat):
        return torch.cat([g.reshape(-1) for g in func(tt, split(flat))])

    sol = odeint(f, torch.cat([y.reshape(-1) for y in y0]), t, **kw)
    return split(sol, (len(t),))


def odeint(func, y0, t, *, method="euler", atol=None, rtol=None, t_cast=True):
    if method not in ("euler", "midpoint", "rk4"):
        raise NotImplementedError(method)
    if isinstance(y0, tuple):
        return _odeint_tuple(func, y0, t, method=method, t_cast=t_cast)

    def f(tt, y):
        return func(tt.to(y.dtype) if t_cast else tt, y)

    # torchdiffeq coerces t to y0's device; dt = t1 - t0 is then a 0-dim tensor on that device, and `dt * k` with a bf16 k
    # rounds dt to bf16 first (type promotion keeps bf16; only a CPU scalar next to a CUDA tensor would stay fp32)
    t = t.to(y0.device)
    sol = torch.empty((len(t),) + tuple(y0.shape), dtype=y0.dtype, device=y0.device)
    sol[0] = y0
    y = y0
    for j in range(len(t) - 1):
        t0, t1 = t[j], t[j + 1]
        dt = t1 - t0
        k1 = f(t0, y)
        if method == "euler":
            dy = dt * k1
        elif method == "midpoint":
            half_dt = 0.5 * dt
            dy = dt * f(t0 + half_dt, y + k1 * half_dt)
        else:
            one_third, two_thirds = 1 / 3, 2 / 3
            k2 = f(t0 + dt * one_third, y + dt * k1 * one_third)
            k3 = f(t0 + dt * two_thirds, y + dt * (k2 - k1 * one_third))
            k4 = f(t1, y + dt * (k1 - k2 + k3))
            dy = (k1 + 3 * (k2 + k3) + k4) * dt * 0.125
        y = y + dy
        sol[j + 1] = y
    return sol


def time_grid(num_steps: int, time_shifting_factor=None, t0: float = 0.0, t1: float = 1.0) -> torch.Tensor:
    """integrators.py:97-99: linspace then t / (t + s - s t)."""
    t = torch.linspace(t0, t1, num_steps)
    if time_shifting_factor:
        s = time_shifting_factor
        t = t / (t + s - s * t)
    return t


def sample_ode(model_fn, z, num_steps, method="euler", time_shifting_factor=None, t_cast=True, **model_kwargs):
    """ode.sample, integrators.py:104-116 with velocity drift (transport.py:181-195)."""
    t = time_grid(num_steps, time_shifting_factor)

    def _fn(tt, x):
        tvec = torch.ones(x.size(0)).to(x.device) * tt  # integrators.py:108
        out = model_fn(x, tvec, **model_kwargs)
        assert out.shape == x.shape
        return out

    return odeint(_fn, z, t, method=method, t_cast=t_cast)
