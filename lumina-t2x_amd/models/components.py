"""Parameter holders whose names/shapes match the reference modules (state_dict compatibility only).

The reference computes in these modules (``lumina_next_t2i/models/components.py:11-54`` RMSNorm, fairscale
Column/RowParallelLinear == nn.Linear at mp=1); here they only carry parameters - all arithmetic of the
denoising path runs in the HIP engine.
"""
import torch
import torch.nn as nn


class RMSNorm(nn.Module):
    """Weight holder for RMSNorm (reference: components.py:11-54); ``eps`` is forwarded to the engine."""

    def __init__(self, dim: int, eps: float = 1e-6):
        super().__init__()
        self.eps = eps
        self.weight = nn.Parameter(torch.ones(dim))


class Linear(nn.Module):
    """``weight [out, in]`` (+ ``bias [out]``) holder, key-compatible with fairscale's parallel linears."""

    def __init__(self, in_features: int, out_features: int, bias: bool = True, init=None):
        super().__init__()
        self.in_features, self.out_features = in_features, out_features
        self.weight = nn.Parameter(torch.empty(out_features, in_features))
        (init or nn.init.xavier_uniform_)(self.weight)
        if bias:
            self.bias = nn.Parameter(torch.zeros(out_features))
        else:
            self.register_parameter("bias", None)


class AffineNorm(nn.Module):
    """``weight``/``bias`` holder for the reference's nn.LayerNorm instances (q_norm, k_norm, ky_norm)."""

    def __init__(self, dim: int, eps: float = 1e-5):
        super().__init__()
        self.eps = eps
        self.weight = nn.Parameter(torch.ones(dim))
        self.bias = nn.Parameter(torch.zeros(dim))
