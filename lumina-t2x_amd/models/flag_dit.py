"""Flag-DiT (Lumina-T2I) behind the reference's construction / checkpoint / call API.

Source compatibility target: ``lumina_t2i/models/model.py`` - ``DiT_Llama(...)``, ``load_state_dict(ckpt, strict=True)``,
``forward(x, t, cap_feats, cap_mask)`` and ``forward_with_cfg(x, t, cap_feats, cap_mask, cfg_scale,
rope_scaling_factor=None, ntk_factor=None, base_seqlen=None, proportional_attn=False)`` (model.py:829-922).
BASELINE configs[2] is ``DiT_Llama_5B_patch2``.  Parameters only; the forward runs on the HIP engine
(variant ``LT_VARIANT_FLAG_T2I``: 1-D RoPE over rows that end in an eol token, shift/scale/gate adaLN, no post-norms).
"""
from __future__ import annotations

from typing import Optional

import torch
import torch.nn as nn

from .. import _lib
from ..engine import ffn_hidden_dim
from ._base import EngineBackedModel
from .components import AffineNorm, Linear, RMSNorm
from .imagenet import FinalLayerShiftScale
from .model import Attention, CapEmbedder, FeedForward, TimestepEmbedder


class TransformerBlock(nn.Module):
    """reference model.py:507-621: keys attention.*, feed_forward.*, attention_norm, ffn_norm, attention_y_norm, adaLN (6 chunks)."""

    def __init__(self, layer_id, dim, n_heads, n_kv_heads, multiple_of, ffn_dim_multiplier, norm_eps, qk_norm, y_dim):
        super().__init__()
        self.dim, self.head_dim, self.layer_id = dim, dim // n_heads, layer_id
        self.attention = Attention(dim, n_heads, n_kv_heads, qk_norm, y_dim)
        self.feed_forward = FeedForward(dim, ffn_hidden_dim(dim, multiple_of, ffn_dim_multiplier))
        self.attention_norm = RMSNorm(dim, eps=norm_eps)
        self.ffn_norm = RMSNorm(dim, eps=norm_eps)
        self.adaLN_modulation = nn.Sequential(nn.SiLU(), Linear(min(dim, 1024), 6 * dim, bias=True, init=nn.init.zeros_))
        self.attention_y_norm = RMSNorm(y_dim, eps=norm_eps)


class DiT_Llama(EngineBackedModel):
    """Constructor signature and defaults follow the reference (model.py:666-682)."""

    _variant = _lib.LT_VARIANT_FLAG_T2I

    def __init__(self, patch_size: int = 2, in_channels: int = 4, dim: int = 4096, n_layers: int = 32, n_heads: int = 32,
                 n_kv_heads: Optional[int] = None, multiple_of: int = 256, ffn_dim_multiplier: Optional[float] = None,
                 norm_eps: float = 1e-5, learn_sigma: bool = True, qk_norm: bool = False, cap_feat_dim: int = 5120,
                 rope_scaling_factor: float = 1.0, ntk_factor: float = 1.0) -> None:
        super().__init__()
        self.learn_sigma, self.in_channels = learn_sigma, in_channels
        self.out_channels = in_channels * 2 if learn_sigma else in_channels
        self.patch_size = patch_size
        self.dim, self.n_heads, self.n_layers = dim, n_heads, n_layers
        self.n_kv_heads = n_heads if n_kv_heads is None else n_kv_heads
        self.norm_eps, self.qk_norm, self.cap_feat_dim = norm_eps, qk_norm, cap_feat_dim
        self.ffn_hidden = ffn_hidden_dim(dim, multiple_of, ffn_dim_multiplier)
        self.rope_scaling_factor, self.ntk_factor = rope_scaling_factor, ntk_factor
        self.x_embedder = Linear(patch_size * patch_size * in_channels, dim, bias=True)
        self.t_embedder = TimestepEmbedder(min(dim, 1024))
        self.cap_embedder = CapEmbedder(AffineNorm(cap_feat_dim), Linear(cap_feat_dim, min(dim, 1024), bias=True, init=nn.init.zeros_))
        self.layers = nn.ModuleList([
            TransformerBlock(i, dim, n_heads, n_kv_heads, multiple_of, ffn_dim_multiplier, norm_eps, qk_norm, cap_feat_dim)
            for i in range(n_layers)])
        self.final_layer = FinalLayerShiftScale(dim, patch_size, self.out_channels)
        self.eol_token = nn.Parameter(torch.empty(dim))
        self.pad_token = nn.Parameter(torch.empty(dim))
        nn.init.normal_(self.eol_token, std=0.02)
        nn.init.normal_(self.pad_token, std=0.02)
        self._init_engine_state()

    def _tokens(self, H: int, W: int) -> int:
        return (H // self.patch_size) * (W // self.patch_size + 1)  # one eol token per latent row (model.py:779-786)

    def _engine_kwargs(self) -> dict:
        return dict(dim=self.dim, n_layers=self.n_layers, n_heads=self.n_heads, n_kv_heads=self.n_kv_heads,
                    ffn_hidden=self.ffn_hidden, patch_size=self.patch_size, in_channels=self.in_channels,
                    out_channels=self.out_channels, cap_feat_dim=self.cap_feat_dim, qk_norm=self.qk_norm, norm_eps=self.norm_eps)

    def _call(self, x, t, cap_feats, cap_mask, use_cfg, **kw):
        if not isinstance(x, torch.Tensor):
            raise NotImplementedError("list-of-latents (variable resolution packing, model.py:789-827) is a later round")
        eng = self.engine(x, cap_feats.shape[1])
        eng.prepare_prompt(cap_feats, cap_mask)
        return eng.forward(x, t, use_cfg=use_cfg, scale_factor=self.rope_scaling_factor, ntk_factor=self.ntk_factor, **kw)

    def _engine_sample_ode(self, x, tgrid, method, use_cfg, t_round, kw):
        """transport fast path (integrators.ode.sample): kwargs of forward_with_cfg / forward -> lt_sample_ode"""
        cap_feats, cap_mask = kw.pop("cap_feats"), kw.pop("cap_mask")
        args = {}
        if use_cfg:
            rs, ntk = kw.pop("rope_scaling_factor", None), kw.pop("ntk_factor", None)
            if rs is not None or ntk is not None:
                self.rope_scaling_factor = rs if rs is not None else self.rope_scaling_factor
                self.ntk_factor = ntk if ntk is not None else self.ntk_factor
            args = dict(cfg_scale=kw.pop("cfg_scale"), base_seqlen=kw.pop("base_seqlen", None),
                        proportional_attn=kw.pop("proportional_attn", False))
        if kw:
            raise TypeError(f"unexpected model kwargs for the engine path: {sorted(kw)}")
        eng = self.engine(x, cap_feats.shape[1])
        eng.prepare_prompt(cap_feats, cap_mask)
        return eng.sample_ode(x, tgrid, method, use_cfg=use_cfg, scale_factor=self.rope_scaling_factor,
                              ntk_factor=self.ntk_factor, t_round_to_state_dtype=t_round, **args)

    @torch.no_grad()
    def forward(self, x, t, cap_feats, cap_mask):
        """reference model.py:829-864"""
        a = self.layers[0].attention if self.n_layers else None
        return self._call(x, t, cap_feats, cap_mask, False, proportional_attn=bool(a and a.proportional_attn),
                          base_seqlen=a.base_seqlen if a else None)

    @torch.no_grad()
    def forward_with_cfg(self, x, t, cap_feats, cap_mask, cfg_scale, rope_scaling_factor=None, ntk_factor=None,
                         base_seqlen: Optional[int] = None, proportional_attn: bool = False):
        """reference model.py:866-922: the RoPE factors persist on the module once overridden (:886-901)."""
        if rope_scaling_factor is not None or ntk_factor is not None:
            self.rope_scaling_factor = rope_scaling_factor if rope_scaling_factor is not None else self.rope_scaling_factor
            self.ntk_factor = ntk_factor if ntk_factor is not None else self.ntk_factor
        if proportional_attn:
            assert base_seqlen is not None
        for layer in self.layers:
            layer.attention.base_seqlen = base_seqlen if proportional_attn else None
            layer.attention.proportional_attn = proportional_attn
        return self._call(x, t, cap_feats, cap_mask, True, cfg_scale=cfg_scale, base_seqlen=base_seqlen,
                          proportional_attn=proportional_attn)


def DiT_Llama_5B_patch2(**kwargs):
    """reference model.py:990-991 (BASELINE configs[2])"""
    return DiT_Llama(patch_size=2, dim=3072, n_layers=32, n_heads=32, **kwargs)
