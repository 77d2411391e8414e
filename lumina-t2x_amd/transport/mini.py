"""Mirror of ``lumina_next_t2i_mini/transport.py`` - the flat ``ODE`` class the mini package (sample.py:146,
sample_img2img.py:146-216) uses instead of Sampler/ode, including the img2img ``strength`` cut of the time grid
(transport.py:79-80).  The model-callable protocol and the engine fast path are those of transport.integrators.ode:
a bound ``forward_with_cfg`` of one of our engine-backed models runs the whole trajectory in one ``lt_sample_ode`` call.

Not mirrored: ``use_sd3=True`` (drives a diffusers SD3Transformer2DModel, not a Lumina model - out of scope)."""
from __future__ import annotations

import torch as th

from .integrators import ADAPTIVE_METHODS, FIXED_GRID_METHODS, _engine_target, adaptive_odeint, fixed_grid_odeint


class ODE:
    """reference lumina_next_t2i_mini/transport.py:57-111"""

    def __init__(self, num_steps, sampler_type="euler", time_shifting_factor=None, t0=0.0, t1=1.0, use_sd3=False,
                 strength=1.0):
        if use_sd3:
            raise NotImplementedError("use_sd3 drives a diffusers SD3 transformer, not a Lumina model (out of scope)")
        self.t = th.linspace(t0, t1, num_steps)
        if time_shifting_factor:
            s = time_shifting_factor
            self.t = self.t / (self.t + s - s * self.t)
        if strength != 1.0:  # img2img: start from the partially noised image at t[int(n (1 - strength))] (transport.py:79-80)
            self.t = self.t[int(num_steps * (1 - strength)):]
        self.use_sd3 = use_sd3
        self.sampler_type = sampler_type
        self.t_round_to_state_dtype = True  # torchdiffeq casts t to the state dtype (see integrators.ode)

    def sample(self, x, model, **model_kwargs):
        if isinstance(x, tuple):
            raise NotImplementedError("tuple states are not part of the sampling path")
        target = _engine_target(model)
        if target is not None and x.is_cuda and self.sampler_type in FIXED_GRID_METHODS and len(self.t) >= 2:
            owner, use_cfg = target
            return owner._engine_sample_ode(x, self.t, self.sampler_type, use_cfg, self.t_round_to_state_dtype,
                                            dict(model_kwargs))
        device = x.device

        def _fn(t, y):
            tvec = th.ones(y.size(0)).to(device) * t  # transport.py:87
            return model(y, tvec, **model_kwargs)

        if self.sampler_type in ADAPTIVE_METHODS:
            return adaptive_odeint(_fn, x, self.t.to(device), method=self.sampler_type)
        return fixed_grid_odeint(_fn, x, self.t.to(device), method=self.sampler_type)
