"""CPU restatement (plain torch, fp32) of the reference Next-DiT forward pass.  TEST INFRASTRUCTURE ONLY.

Every function cites the reference lines it follows (``lumina_next_t2i/models/model.py`` unless noted).
The model is expressed functionally over a reference-format ``state_dict``.

``bf16=True`` additionally applies a bf16 round-trip at every point where the reference's GPU path
(``model.to(bf16)`` under ``torch.autocast``) materialises a bf16 tensor.  That mode is
what the HIP engine is designed to reproduce almost exactly, so it is the sharp debugging oracle; the
fp32 mode is the reference's own CPU path and the parity baseline.
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import torch
import torch.nn.functional as F

from .synth import NextDiTConfig


def _r(x: torch.Tensor, bf16: bool) -> torch.Tensor:
    return x.to(torch.bfloat16).to(torch.float32) if bf16 else x


def timestep_embedding(t: torch.Tensor, dim: int = 256, max_period: float = 10000.0) -> torch.Tensor:
    """model.py:63-82 - cos block first, then sin block."""
    half = dim // 2
    freqs = torch.exp(-math.log(max_period) * torch.arange(0, half, dtype=torch.float32) / half)
    args = t[:, None].float() * freqs[None]
    return torch.cat([torch.cos(args), torch.sin(args)], dim=-1)


def rope_table(head_dim: int, end: int, theta: float = 10000.0, scale_factor: float = 1.0,
               scale_watershed: float = 1.0, timestep: float = 1.0) -> torch.Tensor:
    """model.py:915-963 -> complex64 [end, end, head_dim/2]; slot 2i <- row position, 2i+1 <- column position."""
    if timestep < scale_watershed:
        linear_factor, ntk_factor = scale_factor, 1.0
    else:
        linear_factor, ntk_factor = 1.0, scale_factor
    theta = theta * ntk_factor
    freqs = 1.0 / (theta ** (torch.arange(0, head_dim, 4)[: head_dim // 4].float() / head_dim)) / linear_factor
    ang = torch.outer(torch.arange(end, dtype=torch.float32), freqs).float()
    cis = torch.polar(torch.ones_like(ang), ang)
    nf = head_dim // 4
    cis_h = cis.view(end, 1, nf, 1).repeat(1, end, 1, 1)
    cis_w = cis.view(1, end, nf, 1).repeat(end, 1, 1, 1)
    return torch.cat([cis_h, cis_w], dim=-1).flatten(2)


def apply_rotary(x: torch.Tensor, freqs_cis: torch.Tensor) -> torch.Tensor:
    """model.py:254-282; x [B,N,H,hd] fp32, freqs_cis [1,N,hd/2] complex."""
    xc = torch.view_as_complex(x.float().reshape(*x.shape[:-1], -1, 2))
    return torch.view_as_real(xc * freqs_cis.unsqueeze(2)).flatten(3)


def rmsnorm(x: torch.Tensor, w: torch.Tensor, eps: float, bf16: bool) -> torch.Tensor:
    """components.py:40-54 vanilla RMSNorm: fp32 normalise, cast to x dtype, then multiply by the weight."""
    n = x.float() * torch.rsqrt(x.float().pow(2).mean(-1, keepdim=True) + eps)
    return _r(_r(n, bf16) * w, bf16)


def _linear(x, w, b=None, bf16=False):
    """F.linear; in bf16 mode operands are bf16-valued, accumulation fp32, output rounded to bf16."""
    return _r(F.linear(x, w, b), bf16)


def _sd(sd: Dict[str, torch.Tensor], bf16: bool) -> Dict[str, torch.Tensor]:
    return {k: _r(v.float(), bf16) for k, v in sd.items()}


def attention(sd, p, cfg: NextDiTConfig, x, freqs_cis, y, y_mask, softmax_scale, bf16, region_mask=None):
    """Attention.forward, model.py:337-438 (mask all ones; fp32 SDPA branch :407-418; text branch :420-434).
    ``region_mask`` [Y, N] bool switches on the compositional text branch
    (lumina_next_compositional_generation/models/model.py:422-446)."""
    B, N, _ = x.shape
    H, Hkv, hd = cfg.n_heads, cfg.kv_heads, cfg.head_dim
    xq = _linear(x, sd[p + "wq.weight"], None, bf16)
    xk = _linear(x, sd[p + "wk.weight"], None, bf16)
    xv = _linear(x, sd[p + "wv.weight"], None, bf16)
    if cfg.qk_norm:  # full-width affine LayerNorm, fp32 output under autocast (:361-362)
        xq = F.layer_norm(xq, (H * hd,), sd[p + "q_norm.weight"], sd[p + "q_norm.bias"], 1e-5)
        xk = F.layer_norm(xk, (Hkv * hd,), sd[p + "k_norm.weight"], sd[p + "k_norm.bias"], 1e-5)
    xq = apply_rotary(xq.view(B, N, H, hd), freqs_cis)
    xk = apply_rotary(xk.view(B, N, Hkv, hd), freqs_cis)
    xq, xk = _r(xq, bf16), _r(xk, bf16)  # .to(dtype) (:371)
    xv = xv.view(B, N, Hkv, hd)
    rep = H // Hkv
    kk = xk.repeat_interleave(rep, dim=2) if rep > 1 else xk
    vv = xv.repeat_interleave(rep, dim=2) if rep > 1 else xv
    q_ = xq.permute(0, 2, 1, 3)
    out = F.scaled_dot_product_attention(q_, kk.permute(0, 2, 1, 3), vv.permute(0, 2, 1, 3), scale=softmax_scale)
    out = _r(out.permute(0, 2, 1, 3), bf16)
    if (p + "wk_y.weight") in sd and region_mask is not None:
        # Y captions for ONE image (B = 2 rows): captions 0..Y-2 attend the cond row's queries inside their regions, the
        # last caption the uncond row's; fully masked query rows give NaN -> nan_to_num -> 0; gate; sum over the cond captions
        Y, T = y.shape[0], y.shape[1]
        qy = torch.cat([q_[0:1].expand(Y - 1, -1, -1, -1), q_[-1:]], dim=0)
        yk = _linear(y, sd[p + "wk_y.weight"], None, bf16)
        if cfg.qk_norm:
            yk = F.layer_norm(yk, (Hkv * hd,), sd[p + "ky_norm.weight"], sd[p + "ky_norm.bias"], 1e-5)
        yk = _r(yk, bf16).view(Y, T, Hkv, hd)
        yv = _linear(y, sd[p + "wv_y.weight"], None, bf16).view(Y, T, Hkv, hd)
        if rep > 1:
            yk, yv = yk.repeat_interleave(rep, dim=2), yv.repeat_interleave(rep, dim=2)
        m = y_mask.bool().view(Y, 1, 1, T).expand(Y, H, N, T) & region_mask.view(Y, 1, N, 1)
        oy = F.scaled_dot_product_attention(qy, yk.permute(0, 2, 1, 3), yv.permute(0, 2, 1, 3), m)
        oy = torch.nan_to_num(_r(oy.permute(0, 2, 1, 3), bf16))
        gate = _r(torch.tanh(sd[p + "gate"]), bf16).view(1, 1, -1, 1)
        oy = _r(oy * gate, bf16)
        oy = torch.cat([_r(oy[:-1].sum(dim=0, keepdim=True), bf16), oy[-1:]], dim=0)
        out = _r(out + oy, bf16)
    elif (p + "wk_y.weight") in sd:
        T = y.shape[1]
        yk = _linear(y, sd[p + "wk_y.weight"], None, bf16)
        if cfg.qk_norm:
            yk = F.layer_norm(yk, (Hkv * hd,), sd[p + "ky_norm.weight"], sd[p + "ky_norm.bias"], 1e-5)
        yk = _r(yk, bf16).view(B, T, Hkv, hd)
        yv = _linear(y, sd[p + "wv_y.weight"], None, bf16).view(B, T, Hkv, hd)
        if rep > 1:
            yk, yv = yk.repeat_interleave(rep, dim=2), yv.repeat_interleave(rep, dim=2)
        m = y_mask.bool().view(B, 1, 1, T).expand(B, H, N, T)
        oy = F.scaled_dot_product_attention(q_, yk.permute(0, 2, 1, 3), yv.permute(0, 2, 1, 3), m)
        oy = _r(oy.permute(0, 2, 1, 3), bf16)
        gate = _r(torch.tanh(sd[p + "gate"]), bf16).view(1, 1, -1, 1)
        out = _r(out + _r(oy * gate, bf16), bf16)
    return _linear(out.flatten(-2), sd[p + "wo.weight"], None, bf16)


def feed_forward(sd, p, x, bf16):
    """FeedForward.forward, model.py:497-502."""
    a = _linear(x, sd[p + "w1.weight"], None, bf16)
    b = _linear(x, sd[p + "w3.weight"], None, bf16)
    return _linear(_r(_r(F.silu(a), bf16) * b, bf16), sd[p + "w2.weight"], None, bf16)


def block(sd, i, cfg: NextDiTConfig, x, freqs_cis, y, y_mask, adaln_input, softmax_scale, bf16, region_mask=None):
    """TransformerBlock.forward, model.py:573-624 (adaLN branch)."""
    p = f"layers.{i}."
    mod = _linear(_r(F.silu(adaln_input), bf16), sd[p + "adaLN_modulation.1.weight"], sd[p + "adaLN_modulation.1.bias"], bf16)
    scale_msa, gate_msa, scale_mlp, gate_mlp = mod.chunk(4, dim=1)

    def modulate(v, scale):  # model.py:28-29
        return _r(v * _r(1 + scale.unsqueeze(1), bf16), bf16)

    eps = cfg.norm_eps
    yn = rmsnorm(y, sd[p + "attention_y_norm.weight"], eps, bf16)
    a = attention(sd, p + "attention.", cfg, modulate(rmsnorm(x, sd[p + "attention_norm1.weight"], eps, bf16), scale_msa),
                  freqs_cis, yn, y_mask, softmax_scale, bf16, region_mask)
    x = _r(x + _r(_r(torch.tanh(gate_msa), bf16).unsqueeze(1) * rmsnorm(a, sd[p + "attention_norm2.weight"], eps, bf16), bf16), bf16)
    f = feed_forward(sd, p + "feed_forward.", modulate(rmsnorm(x, sd[p + "ffn_norm1.weight"], eps, bf16), scale_mlp), bf16)
    x = _r(x + _r(_r(torch.tanh(gate_mlp), bf16).unsqueeze(1) * rmsnorm(f, sd[p + "ffn_norm2.weight"], eps, bf16), bf16), bf16)
    return x


def forward(sd_in: Dict[str, torch.Tensor], cfg: NextDiTConfig, x, t, cap_feats, cap_mask, *, freqs_table=None,
            proportional_attn: bool = False, base_seqlen: Optional[int] = None, bf16: bool = False,
            n_layers: Optional[int] = None, return_hidden: bool = False, scale_seqlen: Optional[int] = None,
            regional: Optional[dict] = None):
    """NextDiT.forward, model.py:836-864 (tensor input path :774-788).  ``scale_seqlen`` overrides the sequence length the
    proportional-attention scale is computed from (the padded length of a packed batch, see forward_packed).
    ``regional`` = dict(global_cap_feats [1,Tg,C], global_cap_mask [1,Tg], h_split_num, w_split_num) runs the compositional
    variant (lumina_next_compositional_generation/models/model.py:852-899): cap_feats then holds Y captions for the B = 2 rows."""
    sd = _sd(sd_in, bf16)
    p = cfg.patch_size
    B, C, H, W = x.shape
    hd = cfg.head_dim
    if freqs_table is None:
        freqs_table = rope_table(hd, 384)
    xs = _r(x.float(), bf16)
    tok = xs.view(B, C, H // p, p, W // p, p).permute(0, 2, 4, 1, 3, 5).flatten(3)  # :777
    h = _linear(tok, sd["x_embedder.weight"], sd["x_embedder.bias"], bf16).flatten(1, 2)
    N = h.shape[1]
    freqs_cis = freqs_table[: H // p, : W // p].flatten(0, 1).unsqueeze(0)
    # conditioning (:846-851)
    tf = _r(timestep_embedding(t, 256), bf16)
    te = _linear(tf, sd["t_embedder.mlp.0.weight"], sd["t_embedder.mlp.0.bias"], bf16)
    te = _linear(_r(F.silu(te), bf16), sd["t_embedder.mlp.2.weight"], sd["t_embedder.mlp.2.bias"], bf16)
    cf = _r(cap_feats.float(), bf16)
    region_mask = None
    if regional is None:
        pool_f, mf = cf, cap_mask.float().unsqueeze(-1)
    else:  # pooled conditioning from the global caption (:866-870); region masks (:872-887, the reference's region-id formula)
        pool_f, mf = _r(regional["global_cap_feats"].float(), bf16), regional["global_cap_mask"].float().unsqueeze(-1)
        hs, ws = int(regional["h_split_num"]), int(regional["w_split_num"])
        Y = cap_feats.shape[0]
        rm = torch.zeros(Y, H // p, W // p)
        hps, wps = H // hs // p, W // ws // p
        for i in range(hs):
            for j in range(ws):
                rm[(i + 1) * (j + 1) - 1, hps * i: hps * (i + 1), wps * j: wps * (j + 1)] = 1
        rm[-1] = 1
        region_mask = rm.flatten(1, 2) > 0.5
    pool = _r((pool_f * mf).sum(dim=1) / mf.sum(dim=1), bf16)
    pool = F.layer_norm(pool, (cfg.cap_feat_dim,), sd["cap_embedder.0.weight"], sd["cap_embedder.0.bias"], 1e-5)
    cap_emb = _linear(_r(pool, bf16), sd["cap_embedder.1.weight"], sd["cap_embedder.1.bias"], bf16)
    adaln_input = _r(te + cap_emb, bf16)
    if proportional_attn:
        scale = math.sqrt(math.log(N if scale_seqlen is None else scale_seqlen, base_seqlen) / hd)  # :374
    else:
        scale = math.sqrt(1 / hd)
    L = cfg.n_layers if n_layers is None else n_layers
    hidden = []
    for i in range(L):
        h = block(sd, i, cfg, h, freqs_cis, cf, cap_mask, adaln_input, scale, bf16, region_mask)
        if return_hidden:
            hidden.append(h)
    # final layer (:657-662): affine-free LayerNorm eps 1e-6 in fp32, modulate in fp32, Linear in bf16
    fs = _linear(_r(F.silu(adaln_input), bf16), sd["final_layer.adaLN_modulation.1.weight"],
                 sd["final_layer.adaLN_modulation.1.bias"], bf16)
    hn = F.layer_norm(h, (cfg.dim,), None, None, 1e-6) * _r(1 + fs.unsqueeze(1), bf16)
    o = _linear(_r(hn, bf16), sd["final_layer.linear.weight"], sd["final_layer.linear.bias"], bf16)
    oc = cfg.out_channels
    img = o.view(B, H // p, W // p, p, p, oc).permute(0, 5, 1, 3, 2, 4).flatten(4, 5).flatten(2, 3)  # :753-754
    if cfg.learn_sigma:
        img = img.chunk(2, dim=1)[0]
    return (img, hidden) if return_hidden else img


def forward_packed(sd, cfg: NextDiTConfig, xs, t, cap_feats, cap_mask, *, proportional_attn: bool = False,
                   base_seqlen: Optional[int] = None, bf16: bool = False):
    """NextDiT.forward on a LIST of [C, H_b, W_b] latents (patchify_and_embed list branch, model.py:789-834).
    The reference pads every sequence to the longest one with ``pad_token``, lets padded positions rotate like the sample's
    last token, masks them as keys (:407-415) and drops them in unpatchify (:757-768).  No op mixes tokens except attention,
    where padded keys are masked, so a VALID token's output equals the sample run on its own - with one exception: the
    proportional-attention scale sqrt(log_base(seqlen) / hd) uses the PADDED length (:373-374).  Pinned against the
    reference's own list path in tests/golden/nextdit_tiny_packed.npz."""
    p = cfg.patch_size
    lmax = max((x.shape[1] // p) * (x.shape[2] // p) for x in xs)
    return [forward(sd, cfg, x[None], t[b:b + 1], cap_feats[b:b + 1], cap_mask[b:b + 1], proportional_attn=proportional_attn,
                    base_seqlen=base_seqlen, bf16=bf16, scale_seqlen=lmax)[0] for b, x in enumerate(xs)]


def forward_with_cfg(sd, cfg: NextDiTConfig, x, t, cap_feats, cap_mask, cfg_scale, scale_factor=1.0,
                     scale_watershed=1.0, base_seqlen=None, proportional_attn=False, bf16=False, n_layers=None, regional=None):
    """NextDiT.forward_with_cfg, model.py:866-913 (RoPE table chosen from t[0], CFG on channels [:3]); ``regional``: see forward."""
    table = rope_table(cfg.head_dim, 384, scale_factor=scale_factor, scale_watershed=scale_watershed,
                       timestep=float(t[0]))
    half = x[: len(x) // 2]
    out = forward(sd, cfg, torch.cat([half, half], dim=0), t, cap_feats, cap_mask, freqs_table=table,
                  proportional_attn=proportional_attn, base_seqlen=base_seqlen, bf16=bf16, n_layers=n_layers, regional=regional)
    eps, rest = out[:, :3], out[:, 3:]
    cond, uncond = torch.split(eps, len(eps) // 2, dim=0)
    half_eps = _r(uncond + _r(cfg_scale * _r(cond - uncond, bf16), bf16), bf16)
    return torch.cat([torch.cat([half_eps, half_eps], dim=0), rest], dim=1)


def flops_per_nfe(cfg: NextDiTConfig, n_tokens: int, text_len: int, batch: int = 2) -> float:
    """ALGORITHMIC FLOPs of one forward (2 x MAC; GEMMs + attention) - SURVEY.md 8d formula."""
    d, F_, L, hd = cfg.dim, cfg.ffn_hidden, cfg.n_layers, cfg.head_dim
    dkv = cfg.kv_heads * hd
    per_tok_layer = 2 * (2 * d * d + 2 * d * dkv) + 6 * d * F_ + 4 * n_tokens * d + 4 * text_len * d
    per_txt_layer = 4 * cfg.cap_feat_dim * dkv
    per_sample_layer = 2 * min(d, 1024) * 4 * d
    pp = cfg.patch_size ** 2
    embed = 2 * pp * cfg.in_channels * d + 2 * d * pp * cfg.out_channels
    return float(batch) * (L * (n_tokens * per_tok_layer + text_len * per_txt_layer + per_sample_layer) + n_tokens * embed)
