"""Mirror of ``lumina_next_compositional_generation/models`` - Next-DiT with REGIONAL text cross-attention (one caption per
image region plus the negative / empty caption for the unconditional row; the adaLN conditioning comes from a separate global
caption).  Same parameters and ``state_dict`` keys as ``models.NextDiT``; only the call surface differs
(reference ``models/model.py:852-955``: ``forward`` / ``forward_with_cfg`` take ``global_cap_feats``, ``global_cap_mask``,
``h_split_num``, ``w_split_num``; ``demo.py:208-228`` builds ``cap_feats = [region captions..., negative caption]`` and passes the
joined caption as the global one).  Runs on the HIP engine (``lt_prepare_prompt_regional``); no PyTorch forward."""
from __future__ import annotations

from typing import Optional

import torch

from . import model as _base


class NextDiT(_base.NextDiT):
    def _regional_engine(self, x, cap_feats, cap_mask, global_cap_feats, global_cap_mask, h_split_num, w_split_num):
        if not isinstance(x, torch.Tensor) or x.shape[0] != 2:
            raise ValueError("regional captions: x must be a [2, C, H, W] tensor (cond + uncond row of one image), as in the "
                             "reference's xq[0] / xq[-1] expansion (models/model.py:423)")
        if global_cap_feats is None or global_cap_mask is None:
            raise ValueError("global_cap_feats / global_cap_mask are required (reference models/model.py:866-867)")
        if global_cap_feats.shape[0] != 1:
            raise ValueError("global_cap_feats must hold ONE caption row (demo.py:224: cap_feats[-1:])")
        Y = cap_feats.shape[0]
        lim = self.engine_limits  # caption buffers are sized by max_batch (a NEW object: the engine keeps the old one)
        self.engine_limits = type(lim)(max(lim.max_batch, Y), lim.max_tokens, lim.max_text)
        eng = self.engine(x, max(cap_feats.shape[1], global_cap_feats.shape[1]))
        eng.prepare_prompt_regional(cap_feats, cap_mask, global_cap_feats, global_cap_mask, int(h_split_num), int(w_split_num))
        return eng

    @torch.no_grad()
    def forward(self, x, t, cap_feats, cap_mask, global_cap_feats=None, global_cap_mask=None, h_split_num=1, w_split_num=1):
        """reference models/model.py:852-899"""
        if global_cap_feats is None:  # plain Next-DiT call
            return super().forward(x, t, cap_feats, cap_mask)
        eng = self._regional_engine(x, cap_feats, cap_mask, global_cap_feats, global_cap_mask, h_split_num, w_split_num)
        pa = self.layers[0].attention.proportional_attn if self.n_layers else False
        bs = self.layers[0].attention.base_seqlen if self.n_layers else None
        return eng.forward(x, t, use_cfg=False, scale_factor=self.scale_factor, scale_watershed=0.0, proportional_attn=pa,
                           base_seqlen=bs)

    @torch.no_grad()
    def forward_with_cfg(self, x, t, cap_feats, cap_mask, cfg_scale, scale_factor=1.0, scale_watershed=1.0,
                         base_seqlen: Optional[int] = None, proportional_attn: bool = False, global_cap_feats=None,
                         global_cap_mask=None, h_split_num=1, w_split_num=1):
        """reference models/model.py:901-955"""
        if global_cap_feats is None:
            return super().forward_with_cfg(x, t, cap_feats, cap_mask, cfg_scale, scale_factor, scale_watershed, base_seqlen,
                                            proportional_attn)
        if proportional_attn:
            assert base_seqlen is not None
        for layer in self.layers:
            layer.attention.base_seqlen = base_seqlen if proportional_attn else None
            layer.attention.proportional_attn = proportional_attn
        eng = self._regional_engine(x, cap_feats, cap_mask, global_cap_feats, global_cap_mask, h_split_num, w_split_num)
        return eng.forward(x, t, use_cfg=True, cfg_scale=cfg_scale, scale_factor=scale_factor, scale_watershed=scale_watershed,
                           base_seqlen=base_seqlen, proportional_attn=proportional_attn)

    def _engine_sample_ode(self, x, tgrid, method, use_cfg, t_round, kw):
        if kw.get("global_cap_feats") is None:
            for k in ("global_cap_feats", "global_cap_mask", "h_split_num", "w_split_num"):
                kw.pop(k, None)
            return super()._engine_sample_ode(x, tgrid, method, use_cfg, t_round, kw)
        eng = self._regional_engine(x, kw.pop("cap_feats"), kw.pop("cap_mask"), kw.pop("global_cap_feats"),
                                    kw.pop("global_cap_mask"), kw.pop("h_split_num", 1), kw.pop("w_split_num", 1))
        if use_cfg:
            args = dict(cfg_scale=kw.pop("cfg_scale"), scale_factor=kw.pop("scale_factor", 1.0),
                        scale_watershed=kw.pop("scale_watershed", 1.0), base_seqlen=kw.pop("base_seqlen", None),
                        proportional_attn=kw.pop("proportional_attn", False))
        else:
            args = self._plain_forward_args()
        if kw:
            raise TypeError(f"unexpected model kwargs for the engine path: {sorted(kw)}")
        return eng.sample_ode(x, tgrid, method, use_cfg=use_cfg, t_round_to_state_dtype=t_round, **args)


def NextDiT_2B_patch2(**kwargs):
    return NextDiT(patch_size=2, dim=2304, n_layers=24, n_heads=32, **kwargs)


def NextDiT_2B_GQA_patch2(**kwargs):
    return NextDiT(patch_size=2, dim=2304, n_layers=24, n_heads=32, n_kv_heads=8, **kwargs)
