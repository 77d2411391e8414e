"""Next-DiT (text-conditional) behind the reference's construction / checkpoint / call API.

Source compatibility target: ``lumina_next_t2i/models/model.py`` -
``models.__dict__[name](qk_norm=..., cap_feat_dim=...)`` (sample.py:125-128), ``.eval().to("cuda", dtype)``,
``load_state_dict(ckpt, strict=True)`` with the 567-key contract of SURVEY.md A.2, and
``forward(x, t, cap_feats, cap_mask)`` / ``forward_with_cfg(...)`` (model.py:836-913).

The module tree below only HOLDS parameters under the reference's key names; both call paths hand the
tensors to the HIP engine (``csrc/``) through the C ABI.  There is deliberately no PyTorch implementation of
the forward pass here: on a machine without the HIP extension or without a ROCm device the call raises.
"""
from __future__ import annotations

import functools
from typing import Optional

import torch
import torch.nn as nn

from .. import _lib
from ..engine import DiTEngine, EngineLimits, ffn_hidden_dim
from ._base import WeightWatch
from .components import AffineNorm, Linear, RMSNorm

_normal002 = functools.partial(nn.init.normal_, std=0.02)


class TimestepEmbedder(nn.Module):
    """keys: ``mlp.0.{weight,bias}``, ``mlp.2.{weight,bias}`` (reference model.py:37-60)."""

    def __init__(self, hidden_size: int, frequency_embedding_size: int = 256):
        super().__init__()
        self.mlp = nn.Sequential(
            Linear(frequency_embedding_size, hidden_size, bias=True, init=_normal002),
            nn.SiLU(),
            Linear(hidden_size, hidden_size, bias=True, init=_normal002),
        )
        self.frequency_embedding_size = frequency_embedding_size


class Attention(nn.Module):
    """keys: wq wk wv wo wk_y wv_y (no bias), gate, q_norm/k_norm/ky_norm (reference model.py:137-224)."""

    def __init__(self, dim: int, n_heads: int, n_kv_heads: Optional[int], qk_norm: bool, y_dim: int):
        super().__init__()
        self.n_heads = n_heads
        self.n_kv_heads = n_heads if n_kv_heads is None else n_kv_heads
        self.head_dim = dim // n_heads
        kv_dim = self.n_kv_heads * self.head_dim
        self.wq = Linear(dim, n_heads * self.head_dim, bias=False)
        self.wk = Linear(dim, kv_dim, bias=False)
        self.wv = Linear(dim, kv_dim, bias=False)
        if y_dim > 0:
            self.wk_y = Linear(y_dim, kv_dim, bias=False)
            self.wv_y = Linear(y_dim, kv_dim, bias=False)
            self.gate = nn.Parameter(torch.zeros([n_heads]))
        self.wo = Linear(n_heads * self.head_dim, dim, bias=False)
        if qk_norm:
            self.q_norm = AffineNorm(n_heads * self.head_dim)
            self.k_norm = AffineNorm(kv_dim)
            self.ky_norm = AffineNorm(kv_dim) if y_dim > 0 else nn.Identity()
        else:
            self.q_norm = self.k_norm = nn.Identity()
            self.ky_norm = nn.Identity()
        # kept for source compatibility: forward_with_cfg sets these per call (reference model.py:891-899)
        self.base_seqlen = None
        self.proportional_attn = False


class FeedForward(nn.Module):
    """keys: w1, w2, w3 (reference model.py:441-495)."""

    def __init__(self, dim: int, hidden_dim: int):
        super().__init__()
        self.w1 = Linear(dim, hidden_dim, bias=False)
        self.w2 = Linear(hidden_dim, dim, bias=False)
        self.w3 = Linear(dim, hidden_dim, bias=False)


class TransformerBlock(nn.Module):
    """Sandwich-norm block with tanh-gated adaLN (reference model.py:505-624)."""

    def __init__(self, layer_id: int, dim: int, n_heads: int, n_kv_heads: Optional[int], multiple_of: int,
                 ffn_dim_multiplier: Optional[float], norm_eps: float, qk_norm: bool, y_dim: int):
        super().__init__()
        self.dim = dim
        self.head_dim = dim // n_heads
        self.layer_id = layer_id
        self.attention = Attention(dim, n_heads, n_kv_heads, qk_norm, y_dim)
        self.feed_forward = FeedForward(dim, ffn_hidden_dim(dim, multiple_of, ffn_dim_multiplier))
        self.attention_norm1 = RMSNorm(dim, eps=norm_eps)
        self.ffn_norm1 = RMSNorm(dim, eps=norm_eps)
        self.attention_norm2 = RMSNorm(dim, eps=norm_eps)
        self.ffn_norm2 = RMSNorm(dim, eps=norm_eps)
        self.adaLN_modulation = nn.Sequential(nn.SiLU(), Linear(min(dim, 1024), 4 * dim, bias=True, init=nn.init.zeros_))
        self.attention_y_norm = RMSNorm(y_dim, eps=norm_eps)


class FinalLayer(nn.Module):
    """keys: ``linear.*``, ``adaLN_modulation.1.*`` (reference model.py:627-662; norm_final has no params)."""

    def __init__(self, hidden_size: int, patch_size: int, out_channels: int):
        super().__init__()
        self.linear = Linear(hidden_size, patch_size * patch_size * out_channels, bias=True, init=nn.init.zeros_)
        self.adaLN_modulation = nn.Sequential(
            nn.SiLU(), Linear(min(hidden_size, 1024), hidden_size, bias=True, init=nn.init.zeros_))


class CapEmbedder(nn.Sequential):
    pass


class NextDiT(WeightWatch, nn.Module):
    """Diffusion transformer whose forward passes execute on the MI355X engine.

    Constructor signature and defaults follow the reference (model.py:670-685).  Extra, engine-only knobs
    are keyword-only attributes set after construction: ``engine_limits`` (workspace sizing).
    """

    def __init__(self, patch_size: int = 2, in_channels: int = 4, dim: int = 4096, n_layers: int = 32,
                 n_heads: int = 32, n_kv_heads: Optional[int] = None, multiple_of: int = 256,
                 ffn_dim_multiplier: Optional[float] = None, norm_eps: float = 1e-5, learn_sigma: bool = True,
                 qk_norm: bool = False, cap_feat_dim: int = 5120, scale_factor: float = 1.0,
                 use_flash_attn: bool = True) -> None:
        # use_flash_attn: constructor kwarg of the mini package (lumina_next_t2i_mini/models/nextdit.py:637; sample.py:111) -
        # accepted so its scripts swap in unchanged; the engine has one attention path (exact softmax either way)
        super().__init__()
        self.use_flash_attn = use_flash_attn
        assert (dim // n_heads) % 4 == 0, "2d rope needs head dim to be divisible by 4"
        self.learn_sigma = learn_sigma
        self.in_channels = in_channels
        self.out_channels = in_channels * 2 if learn_sigma else in_channels
        self.patch_size = patch_size
        self.dim, self.n_heads, self.n_layers = dim, n_heads, n_layers
        self.n_kv_heads = n_heads if n_kv_heads is None else n_kv_heads
        self.norm_eps, self.qk_norm, self.cap_feat_dim = norm_eps, qk_norm, cap_feat_dim
        self.scale_factor = scale_factor
        self.ffn_hidden = ffn_hidden_dim(dim, multiple_of, ffn_dim_multiplier)

        self.x_embedder = Linear(patch_size * patch_size * in_channels, dim, bias=True)
        self.t_embedder = TimestepEmbedder(min(dim, 1024))
        cap_ln = AffineNorm(cap_feat_dim)
        self.cap_embedder = CapEmbedder(cap_ln, Linear(cap_feat_dim, min(dim, 1024), bias=True, init=nn.init.zeros_))
        self.layers = nn.ModuleList([
            TransformerBlock(i, dim, n_heads, n_kv_heads, multiple_of, ffn_dim_multiplier, norm_eps, qk_norm, cap_feat_dim)
            for i in range(n_layers)
        ])
        self.final_layer = FinalLayer(dim, patch_size, self.out_channels)
        self.pad_token = nn.Parameter(torch.empty(dim))
        nn.init.normal_(self.pad_token, std=0.02)

        self.engine_limits = EngineLimits()
        self._engine: Optional[DiTEngine] = None
        self._weights_sig = None

    # ---- engine plumbing (the weight-change check is WeightWatch._signature, models/_base.py) -------------------------
    def engine(self, x: torch.Tensor, text_len: int) -> DiTEngine:
        """Create / resize the engine for this call's shapes and make sure it holds the current weights."""
        if not (x if isinstance(x, torch.Tensor) else x[0]).is_cuda:
            raise _lib.LuminaLibError(
                "NextDiT.forward needs tensors on a ROCm device: the MI355X engine has no CPU fallback")
        if isinstance(x, torch.Tensor):
            B, _, H, W = x.shape
            n_tok = (H // self.patch_size) * (W // self.patch_size)
        else:  # list of [C, H_b, W_b] samples: limits of the padded batch
            B = len(x)
            n_tok = max((xi.shape[1] // self.patch_size) * (xi.shape[2] // self.patch_size) for xi in x)
            x = x[0]
        lim = self.engine_limits
        need = EngineLimits(max(lim.max_batch, B), max(lim.max_tokens, n_tok), max(lim.max_text, text_len))
        if self._engine is None or need != self._engine.limits or self._engine.device != x.device:
            self._engine = None
            self._engine = DiTEngine(
                variant=_lib.LT_VARIANT_NEXT_T2I, dim=self.dim, n_layers=self.n_layers, n_heads=self.n_heads,
                n_kv_heads=self.n_kv_heads, ffn_hidden=self.ffn_hidden, patch_size=self.patch_size,
                in_channels=self.in_channels, out_channels=self.out_channels, cap_feat_dim=self.cap_feat_dim,
                qk_norm=self.qk_norm, norm_eps=self.norm_eps, limits=need, device=x.device)
            self.engine_limits = need
            self._weights_sig = None
        sig = self._signature()
        if sig != self._weights_sig:
            self._engine.load_state_dict(self.state_dict())
            self._weights_sig = sig
        return self._engine

    def _call(self, x, t, cap_feats, cap_mask, use_cfg, **kw):
        eng = self.engine(x, cap_feats.shape[1])
        eng.prepare_prompt(cap_feats, cap_mask)
        if not isinstance(x, torch.Tensor):  # variable-resolution packing (model.py:789-834): plain forward only, like the reference
            if use_cfg:
                raise TypeError("forward_with_cfg takes a [B, C, H, W] tensor (reference model.py:901-902)")
            return eng.forward_packed(list(x), t, **kw)
        return eng.forward(x, t, use_cfg=use_cfg, **kw)

    # ---- reference call surface ---------------------------------------------------------------------------
    @torch.no_grad()
    def forward(self, x, t, cap_feats, cap_mask):
        """reference model.py:836-864; ``x`` is a [B, C, H, W] tensor or a list of [C, H_b, W_b] tensors of different sizes
        (returns a list then)"""
        return self._call(x, t, cap_feats, cap_mask, False, **self._plain_forward_args())

    def _plain_forward_args(self) -> dict:
        """what the reference's plain ``forward`` reads off the module (model.py:836-864): the attention flags left on the
        layers by the last forward_with_cfg (:891-899), and the RoPE table left in self.freqs_cis - after construction
        that is the NTK branch at the constructor's scale_factor (:732-736) -> watershed 0 selects it for any t.  (The
        reference also re-uses a table rebuilt by a forward_with_cfg with OTHER scaling arguments; the engine keeps the
        constructor's - documented difference, the reference's samplers never mix the two on one module.)"""
        pa = self.layers[0].attention.proportional_attn if self.n_layers else False
        bs = self.layers[0].attention.base_seqlen if self.n_layers else None
        return dict(scale_factor=self.scale_factor, scale_watershed=0.0, proportional_attn=pa, base_seqlen=bs)

    @torch.no_grad()
    def forward_with_cfg(self, x, t, cap_feats, cap_mask, cfg_scale, scale_factor=1.0, scale_watershed=1.0,
                         base_seqlen: Optional[int] = None, proportional_attn: bool = False):
        """reference model.py:866-913: batch = cat([half, half]); CFG on channels [:3] only."""
        if proportional_attn:
            assert base_seqlen is not None
        for layer in self.layers:  # mirrored attributes (reference model.py:891-899)
            layer.attention.base_seqlen = base_seqlen if proportional_attn else None
            layer.attention.proportional_attn = proportional_attn
        return self._call(x, t, cap_feats, cap_mask, True, cfg_scale=cfg_scale, scale_factor=scale_factor,
                          scale_watershed=scale_watershed, base_seqlen=base_seqlen, proportional_attn=proportional_attn)

    def _engine_sample_ode(self, x, tgrid, method, use_cfg, t_round, kw):
        """transport fast path (integrators.ode.sample): kwargs of forward_with_cfg / forward -> lt_sample_ode"""
        cap_feats, cap_mask = kw.pop("cap_feats"), kw.pop("cap_mask")
        if use_cfg:
            args = dict(cfg_scale=kw.pop("cfg_scale"), scale_factor=kw.pop("scale_factor", 1.0),
                        scale_watershed=kw.pop("scale_watershed", 1.0), base_seqlen=kw.pop("base_seqlen", None),
                        proportional_attn=kw.pop("proportional_attn", False))
        else:
            args = self._plain_forward_args()  # the same module state a per-step forward() would read
        if kw:
            raise TypeError(f"unexpected model kwargs for the engine path: {sorted(kw)}")
        eng = self.engine(x, cap_feats.shape[1])
        eng.prepare_prompt(cap_feats, cap_mask)
        return eng.sample_ode(x, tgrid, method, use_cfg=use_cfg, t_round_to_state_dtype=t_round, **args)

    def parameter_count(self) -> int:
        return sum(p.numel() for p in self.parameters())

    def get_fsdp_wrap_module_list(self):
        return list(self.layers)

    def get_checkpointing_wrap_module_list(self):
        return list(self.layers)


def NextDiT_2B_patch2(**kwargs):
    """reference model.py:994-995"""
    return NextDiT(patch_size=2, dim=2304, n_layers=24, n_heads=32, **kwargs)


def NextDiT_2B_GQA_patch2(**kwargs):
    """reference model.py:998-999"""
    return NextDiT(patch_size=2, dim=2304, n_layers=24, n_heads=32, n_kv_heads=8, **kwargs)
