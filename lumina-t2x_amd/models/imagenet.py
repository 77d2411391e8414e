"""Class-conditional Next-DiT behind the reference's construction / checkpoint / call API.

Source compatibility target: ``Next-DiT-ImageNet/models/models.py`` - ``models.__dict__[name](qk_norm=..., ...)``,
``load_state_dict(ckpt, strict=True)`` and ``forward(x, t, y)`` / ``forward_with_cfg(x, t, y, cfg_scale,
rope_scaling_factor=None, ntk_factor=None)`` (models.py:920-974).  BASELINE configs[0] is ``DiT_Llama_600M_patch2``.
Parameters only; both call paths run on the HIP engine (variant ``LT_VARIANT_NEXT_IMAGENET``).
"""
from __future__ import annotations

import functools
from typing import Optional

import torch
import torch.nn as nn

from .. import _lib
from ..engine import ffn_hidden_dim
from ._base import EngineBackedModel
from .components import AffineNorm, Linear, RMSNorm
from .model import FeedForward, FinalLayer, TimestepEmbedder

_normal002 = functools.partial(nn.init.normal_, std=0.02)


class LabelEmbedder(nn.Module):
    """key: ``embedding_table.weight`` [num_classes + 1, hidden] (reference models.py:182-196; the extra row is the
    null class used for classifier-free guidance)."""

    def __init__(self, num_classes: int, hidden_size: int, dropout_prob: float):
        super().__init__()
        self.embedding_table = nn.Embedding(num_classes + int(dropout_prob > 0), hidden_size)
        _normal002(self.embedding_table.weight)
        self.num_classes, self.dropout_prob = num_classes, dropout_prob


class Attention(nn.Module):
    """keys: wq wk wv wo (no bias), q_norm / k_norm (reference models.py:229-295)."""

    def __init__(self, dim: int, n_heads: int, n_kv_heads: Optional[int], qk_norm: bool):
        super().__init__()
        self.n_heads = n_heads
        self.n_kv_heads = n_heads if n_kv_heads is None else n_kv_heads
        self.head_dim = dim // n_heads
        kv = self.n_kv_heads * self.head_dim
        self.wq, self.wk, self.wv = Linear(dim, dim, bias=False), Linear(dim, kv, bias=False), Linear(dim, kv, bias=False)
        self.wo = Linear(dim, dim, bias=False)
        self.q_norm = AffineNorm(dim) if qk_norm else nn.Identity()
        self.k_norm = AffineNorm(kv) if qk_norm else nn.Identity()


class TransformerBlockSandwichNorm2(nn.Module):
    """reference models.py:692-796: weight-free pre-norms (PFRMSNorm), weighted post-norms, tanh-gated adaLN (4 chunks)."""

    def __init__(self, layer_id, dim, n_heads, n_kv_heads, multiple_of, ffn_dim_multiplier, norm_eps, qk_norm):
        super().__init__()
        self.dim, self.head_dim, self.layer_id = dim, dim // n_heads, layer_id
        self.attention = Attention(dim, n_heads, n_kv_heads, qk_norm)
        self.feed_forward = FeedForward(dim, ffn_hidden_dim(dim, multiple_of, ffn_dim_multiplier))
        self.attention_norm = RMSNorm(dim, eps=norm_eps)
        self.ffn_norm = RMSNorm(dim, eps=norm_eps)
        self.adaLN_modulation = nn.Sequential(nn.SiLU(), Linear(min(dim, 1024), 4 * dim, bias=True, init=nn.init.zeros_))


class FinalLayerShiftScale(FinalLayer):
    """reference models.py:799-833: adaLN produces (shift, scale) -> 2 * hidden rows"""

    def __init__(self, hidden_size: int, patch_size: int, out_channels: int):
        super().__init__(hidden_size, patch_size, out_channels)
        self.adaLN_modulation = nn.Sequential(
            nn.SiLU(), Linear(min(hidden_size, 1024), 2 * hidden_size, bias=True, init=nn.init.zeros_))


class DiT_Llama(EngineBackedModel):
    """Constructor signature and defaults follow the reference (models.py:841-857)."""

    _variant = _lib.LT_VARIANT_NEXT_IMAGENET

    def __init__(self, input_size: int = 32, patch_size: int = 2, in_channels: int = 4, dim: int = 4096, n_layers: int = 32,
                 n_heads: int = 32, n_kv_heads: Optional[int] = None, multiple_of: int = 256,
                 ffn_dim_multiplier: Optional[float] = None, norm_eps: float = 1e-5, class_dropout_prob: float = 0.1,
                 num_classes: int = 1000, learn_sigma: bool = True, qk_norm: bool = False) -> None:
        super().__init__()
        assert (dim // n_heads) % 4 == 0, "2d rope needs head dim to be divisible by 4"
        self.learn_sigma, self.in_channels = learn_sigma, in_channels
        self.out_channels = in_channels * 2 if learn_sigma else in_channels
        self.input_size, self.patch_size = input_size, patch_size
        self.dim, self.n_heads, self.n_layers = dim, n_heads, n_layers
        self.n_kv_heads = n_heads if n_kv_heads is None else n_kv_heads
        self.norm_eps, self.qk_norm, self.num_classes = norm_eps, qk_norm, num_classes
        self.ffn_hidden = ffn_hidden_dim(dim, multiple_of, ffn_dim_multiplier)
        self.x_embedder = Linear(patch_size * patch_size * in_channels, dim, bias=True)
        self.t_embedder = TimestepEmbedder(min(dim, 1024))
        self.y_embedder = LabelEmbedder(num_classes, min(dim, 1024), class_dropout_prob)
        self.layers = nn.ModuleList([
            TransformerBlockSandwichNorm2(i, dim, n_heads, n_kv_heads, multiple_of, ffn_dim_multiplier, norm_eps, qk_norm)
            for i in range(n_layers)])
        self.final_layer = FinalLayerShiftScale(dim, patch_size, self.out_channels)
        # forward_with_cfg(rope_scaling_factor=..., ntk_factor=...) overrides the table for all later calls (models.py:952-956)
        self._rope = (1.0, 1.0)
        self._init_engine_state()

    def _engine_kwargs(self) -> dict:
        return dict(dim=self.dim, n_layers=self.n_layers, n_heads=self.n_heads, n_kv_heads=self.n_kv_heads,
                    ffn_hidden=self.ffn_hidden, patch_size=self.patch_size, in_channels=self.in_channels,
                    out_channels=self.out_channels, cap_feat_dim=0, qk_norm=self.qk_norm, norm_eps=self.norm_eps,
                    num_classes=self.y_embedder.embedding_table.weight.shape[0] - 1)

    def _call(self, x, t, y, use_cfg, cfg_scale=1.0):
        eng = self.engine(x)
        eng.prepare_labels(y)
        return eng.forward(x, t, use_cfg=use_cfg, cfg_scale=cfg_scale, scale_factor=self._rope[0], ntk_factor=self._rope[1])

    def _engine_sample_ode(self, x, tgrid, method, use_cfg, t_round, kw):
        """transport fast path (integrators.ode.sample): kwargs of forward_with_cfg / forward -> lt_sample_ode"""
        y = kw.pop("y")
        cfg_scale = kw.pop("cfg_scale") if use_cfg else 1.0
        rs, ntk = kw.pop("rope_scaling_factor", None), kw.pop("ntk_factor", None)
        if kw:
            raise TypeError(f"unexpected model kwargs for the engine path: {sorted(kw)}")
        if rs is not None or ntk is not None:
            assert rs is not None and ntk is not None
            self._rope = (float(rs), float(ntk))
        eng = self.engine(x)
        eng.prepare_labels(y)
        return eng.sample_ode(x, tgrid, method, use_cfg=use_cfg, cfg_scale=cfg_scale, scale_factor=self._rope[0],
                              ntk_factor=self._rope[1], t_round_to_state_dtype=t_round)

    @torch.no_grad()
    def forward(self, x, t, y):
        """reference models.py:920-944"""
        return self._call(x, t, y, False)

    @torch.no_grad()
    def forward_with_cfg(self, x, t, y, cfg_scale, rope_scaling_factor=None, ntk_factor=None):
        """reference models.py:946-974: batch = cat([half, half]); CFG on channels [:3] only."""
        if rope_scaling_factor is not None or ntk_factor is not None:
            assert rope_scaling_factor is not None and ntk_factor is not None
            self._rope = (float(rope_scaling_factor), float(ntk_factor))
        return self._call(x, t, y, True, cfg_scale)


def DiT_Llama_600M_patch2(**kwargs):
    """reference models.py:1042-1043 (BASELINE configs[0])"""
    return DiT_Llama(patch_size=2, dim=1536, n_layers=16, n_heads=32, **kwargs)


def DiT_Llama_2B_patch2(**kwargs):
    return DiT_Llama(patch_size=2, dim=2304, n_layers=24, n_heads=32, **kwargs)


def DiT_Llama_3B_patch2(**kwargs):
    return DiT_Llama(patch_size=2, dim=3072, n_layers=32, n_heads=32, **kwargs)


def DiT_Llama_7B_patch2(**kwargs):
    return DiT_Llama(patch_size=2, dim=4096, n_layers=32, n_heads=32, **kwargs)
