"""Probability paths x_t = alpha_t x1 + sigma_t x0 used by the transport (flow-matching) formulation.

API mirror of ``lumina_next_t2i/transport/path.py`` (ICPlan :19-144, VPCPlan :147-185, GVPCPlan :188-201):
same class and method names, same return conventions.  Each plan only supplies its schedule
(alpha, alpha', sigma, sigma'); every derived quantity is computed once in the base class.
"""
import math

import torch as th


def expand_t_like_x(t, x):
    """[B] time vector -> [B, 1, 1, ...] broadcastable against x (x may be a list of per-sample tensors)."""
    ref = x[0]
    return t.reshape(t.size(0), *([1] * ref.dim()))


class ICPlan:
    """Linear ("rectified flow") coupling: alpha_t = t, sigma_t = 1 - t."""

    def __init__(self, sigma=0.0):
        self.sigma = sigma

    # ---- schedule ---------------------------------------------------------------------------------
    def compute_alpha_t(self, t):
        return t, 1

    def compute_sigma_t(self, t):
        return 1 - t, -1

    def compute_d_alpha_alpha_ratio_t(self, t):
        return 1 / t

    # ---- derived ------------------------------------------------------------------------------------
    def compute_drift(self, x, t):
        """(-f(x,t), g^2/2 - like diffusion term) of the score-parametrised SDE"""
        t = expand_t_like_x(t, x)
        ratio = self.compute_d_alpha_alpha_ratio_t(t)
        sigma_t, d_sigma_t = self.compute_sigma_t(t)
        return -(ratio * x), ratio * (sigma_t**2) - sigma_t * d_sigma_t

    def compute_diffusion(self, x, t, form="constant", norm=1.0):
        t = expand_t_like_x(t, x)
        forms = {
            "constant": lambda: norm,
            "SBDM": lambda: norm * self.compute_drift(x, t)[1],
            "sigma": lambda: norm * self.compute_sigma_t(t)[0],
            "linear": lambda: norm * (1 - t),
            "decreasing": lambda: 0.25 * (norm * th.cos(math.pi * t) + 1) ** 2,
            "inccreasing-decreasing": lambda: norm * th.sin(math.pi * t) ** 2,
        }
        if form not in forms:
            raise NotImplementedError(f"Diffusion form {form} not implemented")
        return forms[form]()

    def _ratio_and_coeffs(self, x, t):
        t = expand_t_like_x(t, x)
        alpha_t, d_alpha_t = self.compute_alpha_t(t)
        sigma_t, d_sigma_t = self.compute_sigma_t(t)
        return alpha_t / d_alpha_t, sigma_t, d_sigma_t

    def get_score_from_velocity(self, velocity, x, t):
        r, sigma_t, d_sigma_t = self._ratio_and_coeffs(x, t)
        var = sigma_t**2 - r * d_sigma_t * sigma_t
        return (r * velocity - x) / var

    def get_noise_from_velocity(self, velocity, x, t):
        r, sigma_t, d_sigma_t = self._ratio_and_coeffs(x, t)
        var = r * d_sigma_t - sigma_t
        return (r * velocity - x) / var

    def get_velocity_from_score(self, score, x, t):
        drift, var = self.compute_drift(x, t)
        return var * score - drift

    # ---- training-side helpers ------------------------------------------------------------------------
    def compute_mu_t(self, t, x0, x1):
        t = expand_t_like_x(t, x1)
        alpha_t, _ = self.compute_alpha_t(t)
        sigma_t, _ = self.compute_sigma_t(t)
        if isinstance(x1, (list, tuple)):
            return [alpha_t[i] * x1[i] + sigma_t[i] * x0[i] for i in range(len(x1))]
        return alpha_t * x1 + sigma_t * x0

    def compute_xt(self, t, x0, x1):
        return self.compute_mu_t(t, x0, x1)

    def compute_ut(self, t, x0, x1, xt):
        t = expand_t_like_x(t, x1)
        _, d_alpha_t = self.compute_alpha_t(t)
        _, d_sigma_t = self.compute_sigma_t(t)
        if isinstance(x1, (list, tuple)):
            return [d_alpha_t * x1[i] + d_sigma_t * x0[i] for i in range(len(x1))]
        return d_alpha_t * x1 + d_sigma_t * x0

    def plan(self, t, x0, x1):
        xt = self.compute_xt(t, x0, x1)
        return t, xt, self.compute_ut(t, x0, x1, xt)


class VPCPlan(ICPlan):
    """Variance-preserving path with a linear beta schedule between sigma_min and sigma_max."""

    def __init__(self, sigma_min=0.1, sigma_max=20.0):
        self.sigma_min, self.sigma_max = sigma_min, sigma_max
        span = sigma_max - sigma_min
        self.log_mean_coeff = lambda t: -0.25 * ((1 - t) ** 2) * span - 0.5 * (1 - t) * sigma_min
        self.d_log_mean_coeff = lambda t: 0.5 * (1 - t) * span + 0.5 * sigma_min

    def compute_alpha_t(self, t):
        alpha_t = th.exp(self.log_mean_coeff(t))
        return alpha_t, alpha_t * self.d_log_mean_coeff(t)

    def compute_sigma_t(self, t):
        e2 = th.exp(2 * self.log_mean_coeff(t))
        sigma_t = th.sqrt(1 - e2)
        return sigma_t, e2 * (2 * self.d_log_mean_coeff(t)) / (-2 * sigma_t)

    def compute_d_alpha_alpha_ratio_t(self, t):
        return self.d_log_mean_coeff(t)

    def compute_drift(self, x, t):
        t = expand_t_like_x(t, x)
        beta_t = self.sigma_min + (1 - t) * (self.sigma_max - self.sigma_min)
        return -0.5 * beta_t * x, beta_t / 2


class GVPCPlan(ICPlan):
    """Trigonometric ("generalised VP") path: alpha = sin(pi t / 2), sigma = cos(pi t / 2)."""

    def compute_alpha_t(self, t):
        return th.sin(t * math.pi / 2), math.pi / 2 * th.cos(t * math.pi / 2)

    def compute_sigma_t(self, t):
        return th.cos(t * math.pi / 2), -math.pi / 2 * th.sin(t * math.pi / 2)

    def compute_d_alpha_alpha_ratio_t(self, t):
        return math.pi / (2 * th.tan(t * math.pi / 2))
