"""CPU restatement (plain torch, fp32) of the OTHER model families on the denoising path.  TEST INFRASTRUCTURE ONLY.

  imagenet  Next-DiT-ImageNet/models/models.py  DiT_Llama   class-conditional Next-DiT (BASELINE configs[0])
  flag_t2i  lumina_t2i/models/model.py           DiT_Llama   Flag-DiT, text-conditional (BASELINE configs[2])
  moe       Next-DiT-MoE/models/models2.py       DiT_Llama   Next-DiT with time + space MoE FFNs (BASELINE configs[4])

Each function cites the reference lines it follows.  Shared pieces (timestep embedding, RMSNorm, Attention incl. the
gated text branch, FeedForward, 2-D RoPE) come from ``nextdit_oracle`` - the reference copies them verbatim between
its sub-projects.  Pinned like ``nextdit_oracle``: ``oracle/make_golden.py`` runs the unmodified reference modules
behind the stubs and ``tests/test_oracle_golden.py`` compares (tests/golden/{imagenet,flag,moe}_tiny.npz).
``bf16=True`` applies the reference's bf16 rounding points (model in bf16, fp32 norms / RoPE / softmax).
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import torch
import torch.nn.functional as F

from .nextdit_oracle import _linear, _r, _sd, apply_rotary, attention, feed_forward, rmsnorm, timestep_embedding
from .synth import NextDiTConfig


def pf_rmsnorm(x: torch.Tensor, eps: float, bf16: bool) -> torch.Tensor:
    """PFRMSNorm (Next-DiT-ImageNet/models/models.py:76-117): RMSNorm without a weight, cast back to x's dtype."""
    return _r(x.float() * torch.rsqrt(x.float().pow(2).mean(-1, keepdim=True) + eps), bf16)


def rope_table_2d_general(head_dim: int, end: int, theta: float = 10000.0, rope_scaling_factor: float = 1.0,
                          ntk_factor: float = 1.0) -> torch.Tensor:
    """DiT_Llama.precompute_freqs_cis of the ImageNet / MoE sub-projects (models.py:977-1012): both factors apply at
    once (no watershed).  complex64 [end, end, head_dim/2]; slot 2i <- row position, slot 2i+1 <- column position."""
    theta = theta * ntk_factor
    freqs = 1.0 / (theta ** (torch.arange(0, head_dim, 4)[: head_dim // 4].float() / head_dim))
    t = torch.arange(end, dtype=torch.float32) / rope_scaling_factor
    ang = torch.outer(t, freqs).float()
    cis = torch.polar(torch.ones_like(ang), ang)
    nf = head_dim // 4
    cis_h = cis.view(end, 1, nf, 1).repeat(1, end, 1, 1)
    cis_w = cis.view(1, end, nf, 1).repeat(end, 1, 1, 1)
    return torch.cat([cis_h, cis_w], dim=-1).flatten(2)


def rope_table_1d(head_dim: int, end: int, theta: float = 10000.0, rope_scaling_factor: float = 1.0,
                  ntk_factor: float = 1.0) -> torch.Tensor:
    """Flag-DiT precompute_freqs_cis (lumina_t2i/models/model.py:924-960): complex64 [end, head_dim/2], one position
    per token of the flattened sequence (eol tokens included)."""
    theta = theta * ntk_factor
    freqs = 1.0 / (theta ** (torch.arange(0, head_dim, 2)[: head_dim // 2].float() / head_dim))
    t = torch.arange(end, dtype=torch.float32) / rope_scaling_factor
    ang = torch.outer(t, freqs).float()
    return torch.polar(torch.ones_like(ang), ang)


def _t_embed(sd, t, bf16):
    """ParallelTimestepEmbedder.forward (models.py:176-179 / model.py:84-87)."""
    tf = _r(timestep_embedding(t, 256), bf16)
    te = _linear(tf, sd["t_embedder.mlp.0.weight"], sd["t_embedder.mlp.0.bias"], bf16)
    return _linear(_r(F.silu(te), bf16), sd["t_embedder.mlp.2.weight"], sd["t_embedder.mlp.2.bias"], bf16)


def _final_shift_scale(sd, cfg, h, adaln_input, bf16):
    """ParallelFinalLayer.forward with (shift, scale) (models.py:829-833 / model.py:654-658): affine-free LayerNorm
    eps 1e-6 (fp32 under autocast), modulate, Linear."""
    fs = _linear(_r(F.silu(adaln_input), bf16), sd["final_layer.adaLN_modulation.1.weight"],
                 sd["final_layer.adaLN_modulation.1.bias"], bf16)
    shift, scale = fs.chunk(2, dim=1)
    hn = F.layer_norm(h, (cfg.dim,), None, None, 1e-6) * _r(1 + scale.unsqueeze(1), bf16) + shift.unsqueeze(1)
    return _linear(_r(hn, bf16), sd["final_layer.linear.weight"], sd["final_layer.linear.bias"], bf16)


def _cfg_combine(out, cfg_scale, bf16):
    """CFG on channels [:3] only (models.py:965-974 / model.py:913-922)."""
    eps, rest = out[:, :3], out[:, 3:]
    cond, uncond = torch.split(eps, len(eps) // 2, dim=0)
    half_eps = _r(uncond + _r(cfg_scale * _r(cond - uncond, bf16), bf16), bf16)
    return torch.cat([torch.cat([half_eps, half_eps], dim=0), rest], dim=1)


# ---- time / space MoE (Next-DiT-MoE/models/models2.py:451-506) ---------------------------------------------------------

class MoeRouting:
    """parity instrument, not part of the reference: records the experts every MoE layer selects ([L][2 branches][rows][2],
    ascending ids per row, -1 where a branch does not run) and / or replaces the discrete top-2 choice with a given table while
    the softmax weights stay the run's own arithmetic (what lt_moe_routing_record / _force do in the engine)."""

    def __init__(self, n_layers, force=None):
        self.n_layers, self.force, self.rec = n_layers, force, {}

    def table(self):
        import numpy as np
        rows = max(v.shape[0] for v in self.rec.values())
        out = np.full((self.n_layers, 2, rows, 2), -1, dtype=np.int32)
        for (l, b), v in self.rec.items():
            out[l, b, : v.shape[0]] = v
        return out


def _moe_combine(x2d, logits, experts_fn, n_experts, top_k, bf16, hook=None, where=None):
    """shared tail of TimeMoeLayer / SpaceMoeLayer.forward (:464-477 / :493-506): top-k over the router logits, fp32
    softmax over the selected logits, cast to the activation dtype, then for expert 0, 1, ... in order
    ``results[rows] += weight * expert(rows)`` (results starts as zeros in the activation dtype)."""
    weights, selected = torch.topk(logits, top_k)
    if hook is not None:
        if hook.force is not None:
            selected = torch.from_numpy(hook.force[where[0], where[1], : logits.shape[0]]).long()
            weights = torch.gather(logits, 1, selected)
        hook.rec[where] = torch.sort(selected, dim=1).values.to(torch.int32).numpy().copy()
    weights = _r(F.softmax(weights.float(), dim=1), bf16)
    results = torch.zeros_like(x2d)
    for e in range(n_experts):
        rows, nth = torch.where(selected == e)
        if rows.numel() == 0:
            continue
        results[rows] = _r(results[rows] + _r(weights[rows, nth, None] * experts_fn(e, x2d[rows]), bf16), bf16)
    return results


def moe_time(sd, p, cfg, x, cond, bf16, hook=None, layer=0):
    """TimeMoeLayer.forward (:459-477): the router sees the TIME embedding, so every token of a sample shares experts."""
    B, N, d = x.shape
    logits = _linear(cond, sd[p + "gate.weight"], None, bf16)              # nn.Linear(min(dim,1024), E, bias=False)
    logits = logits.repeat(1, N).view(B * N, -1)
    out = _moe_combine(x.reshape(B * N, d), logits, lambda e, rows: feed_forward(sd, p + f"experts.{e}.", rows, bf16),
                       cfg.num_experts, cfg.num_experts_per_tok, bf16, hook, (layer, 0))
    return out.view(B, N, d)


def moe_space(sd, p, cfg, x, bf16, hook=None, layer=0):
    """SpaceMoeLayer.forward (:488-506): per-token router on the FFN input."""
    B, N, d = x.shape
    x2 = x.reshape(B * N, d)
    logits = _linear(x2, sd[p + "gate.weight"], None, bf16)               # nn.Linear(dim, E, bias=False)
    out = _moe_combine(x2, logits, lambda e, rows: feed_forward(sd, p + f"experts.{e}.", rows, bf16),
                       cfg.num_experts, cfg.num_experts_per_tok, bf16, hook, (layer, 1))
    return out.view(B, N, d)


# ---- class-conditional Next-DiT (ImageNet) and its MoE sibling ----------------------------------------------------------

def imagenet_block(sd, i, cfg: NextDiTConfig, x, freqs_cis, adaln_input, time_input, bf16, moe_hook=None):
    """TransformerBlockSandwichNorm2.forward: Next-DiT-ImageNet/models/models.py:759-796 (4 chunks) and
    Next-DiT-MoE/models/models2.py:769-820 (6 chunks, two FFN branches, the second on the updated stream)."""
    p = f"layers.{i}."
    eps = cfg.norm_eps
    mod = _linear(_r(F.silu(adaln_input), bf16), sd[p + "adaLN_modulation.1.weight"], sd[p + "adaLN_modulation.1.bias"], bf16)
    ch = mod.chunk(cfg.chunks, dim=1)

    def mod1(v, scale):
        return _r(v * _r(1 + scale.unsqueeze(1), bf16), bf16)

    def gated(xres, gate, branch, wkey):
        return _r(xres + _r(_r(torch.tanh(gate), bf16).unsqueeze(1) * rmsnorm(branch, sd[p + wkey], eps, bf16), bf16), bf16)

    scale = math.sqrt(1 / cfg.head_dim)  # flash_attn_func default softmax_scale (models.py:389)
    a = attention(sd, p + "attention.", cfg, mod1(pf_rmsnorm(x, eps, bf16), ch[0]), freqs_cis, None, None, scale, bf16)
    h = gated(x, ch[1], a, "attention_norm.weight")
    if cfg.family == "imagenet":
        f = feed_forward(sd, p + "feed_forward.", mod1(pf_rmsnorm(h, eps, bf16), ch[2]), bf16)
        return gated(h, ch[3], f, "ffn_norm.weight")
    if cfg.family == "moe_time":   # Next-DiT-MoE/models/models.py:755-758: `feed_forward` is ONE time-routed MoeLayer
        f = moe_time(sd, p + "feed_forward.", cfg, mod1(pf_rmsnorm(h, eps, bf16), ch[2]), time_input, bf16, moe_hook, i)
        return gated(h, ch[3], f, "ffn_norm.weight")
    if cfg.family == "moe_space":  # Next-DiT-MoE/models/models1.py:755-758: ONE token-routed MoeLayer
        f = moe_space(sd, p + "feed_forward.", cfg, mod1(pf_rmsnorm(h, eps, bf16), ch[2]), bf16, moe_hook, i)
        return gated(h, ch[3], f, "ffn_norm.weight")
    ft = moe_time(sd, p + "feed_forward_time.", cfg, mod1(pf_rmsnorm(h, eps, bf16), ch[2]), time_input, bf16, moe_hook, i)
    h = gated(h, ch[3], ft, "ffn_norm_time.weight")
    fs = moe_space(sd, p + "feed_forward_space.", cfg, mod1(pf_rmsnorm(h, eps, bf16), ch[4]), bf16, moe_hook, i)
    return gated(h, ch[5], fs, "ffn_norm_space.weight")


def imagenet_forward(sd_in: Dict[str, torch.Tensor], cfg: NextDiTConfig, x, t, y, *, freqs_table=None, bf16: bool = False,
                     n_layers: Optional[int] = None, return_hidden: bool = False, moe_hook=None):
    """DiT_Llama.forward (Next-DiT-ImageNet/models/models.py:920-944; MoE: models2.py:930-958, which hands the
    timestep embedding to every block as ``time_input``)."""
    sd = _sd(sd_in, bf16)
    p = cfg.patch_size
    B, C, H, W = x.shape
    if freqs_table is None:
        freqs_table = rope_table_2d_general(cfg.head_dim, 384)
    xs = _r(x.float(), bf16)
    tok = xs.view(B, C, H // p, p, W // p, p).permute(0, 2, 4, 1, 3, 5).flatten(-3).flatten(1, 2)  # patchify :911-918
    h = _linear(tok, sd["x_embedder.weight"], sd["x_embedder.bias"], bf16)
    freqs_cis = freqs_table[: H // p, : W // p].flatten(0, 1).unsqueeze(0)
    te = _t_embed(sd, t, bf16)
    ye = sd["y_embedder.embedding_table.weight"][y.long()]  # eval: no label dropout (models.py:216-221)
    adaln_input = _r(te + ye, bf16)
    L = cfg.n_layers if n_layers is None else n_layers
    hidden = []
    for i in range(L):
        h = imagenet_block(sd, i, cfg, h, freqs_cis, adaln_input, te, bf16, moe_hook)
        if return_hidden:
            hidden.append(h)
    o = _final_shift_scale(sd, cfg, h, adaln_input, bf16)
    oc = cfg.out_channels
    img = o.reshape(B, H // p, W // p, p, p, oc)
    img = torch.einsum("nhwpqc->nchpwq", img).reshape(B, oc, H, W)  # unpatchify :897-909
    if cfg.learn_sigma:
        img = img.chunk(2, dim=1)[0]
    return (img, hidden) if return_hidden else img


def imagenet_forward_with_cfg(sd, cfg: NextDiTConfig, x, t, y, cfg_scale, rope_scaling_factor=None, ntk_factor=None,
                              bf16=False, n_layers=None, moe_hook=None):
    """DiT_Llama.forward_with_cfg (models.py:946-974)."""
    table = None
    if rope_scaling_factor is not None or ntk_factor is not None:
        assert rope_scaling_factor is not None and ntk_factor is not None
        table = rope_table_2d_general(cfg.head_dim, 384, rope_scaling_factor=rope_scaling_factor, ntk_factor=ntk_factor)
    half = x[: len(x) // 2]
    out = imagenet_forward(sd, cfg, torch.cat([half, half], dim=0), t, y, freqs_table=table, bf16=bf16, n_layers=n_layers,
                           moe_hook=moe_hook)
    return _cfg_combine(out, cfg_scale, bf16)


# ---- Flag-DiT (lumina_t2i) ------------------------------------------------------------------------------------------------

def flag_block(sd, i, cfg: NextDiTConfig, x, freqs_cis, y, y_mask, adaln_input, softmax_scale, bf16):
    """TransformerBlock.forward (lumina_t2i/models/model.py:572-621): pre-norm only, shift/scale/gate, plain gate."""
    p = f"layers.{i}."
    eps = cfg.norm_eps
    mod = _linear(_r(F.silu(adaln_input), bf16), sd[p + "adaLN_modulation.1.weight"], sd[p + "adaLN_modulation.1.bias"], bf16)
    shift_msa, scale_msa, gate_msa, shift_mlp, scale_mlp, gate_mlp = mod.chunk(6, dim=1)

    def modulate(v, shift, scale):  # model.py:28-29
        return _r(_r(v * _r(1 + scale.unsqueeze(1), bf16), bf16) + shift.unsqueeze(1), bf16)

    yn = rmsnorm(y, sd[p + "attention_y_norm.weight"], eps, bf16)
    a = attention(sd, p + "attention.", cfg, modulate(rmsnorm(x, sd[p + "attention_norm.weight"], eps, bf16), shift_msa, scale_msa),
                  freqs_cis, yn, y_mask, softmax_scale, bf16)
    x = _r(x + _r(gate_msa.unsqueeze(1) * a, bf16), bf16)
    f = feed_forward(sd, p + "feed_forward.", modulate(rmsnorm(x, sd[p + "ffn_norm.weight"], eps, bf16), shift_mlp, scale_mlp), bf16)
    return _r(x + _r(gate_mlp.unsqueeze(1) * f, bf16), bf16)


def flag_forward(sd_in: Dict[str, torch.Tensor], cfg: NextDiTConfig, x, t, cap_feats, cap_mask, *, freqs_table=None,
                 proportional_attn: bool = False, base_seqlen: Optional[int] = None, bf16: bool = False,
                 n_layers: Optional[int] = None, return_hidden: bool = False):
    """DiT_Llama.forward (model.py:829-864), tensor input path of patchify_and_embed (:774-788: one eol token per row)."""
    sd = _sd(sd_in, bf16)
    p = cfg.patch_size
    B, C, H, W = x.shape
    Hp, Wp = H // p, W // p
    hd = cfg.head_dim
    xs = _r(x.float(), bf16)
    tok = xs.view(B, C, Hp, p, Wp, p).permute(0, 2, 4, 1, 3, 5).flatten(3)
    h = _linear(tok, sd["x_embedder.weight"], sd["x_embedder.bias"], bf16)
    h = torch.cat([h, sd["eol_token"].view(1, 1, 1, -1).expand(B, Hp, 1, -1)], dim=2).flatten(1, 2)
    N = h.shape[1]
    if freqs_table is None:
        freqs_table = rope_table_1d(hd, N)
    freqs_cis = freqs_table[:N].unsqueeze(0)
    te = _t_embed(sd, t, bf16)
    cf = _r(cap_feats.float(), bf16)
    mf = cap_mask.float().unsqueeze(-1)
    pool = _r((cf * mf).sum(dim=1) / mf.sum(dim=1), bf16)
    pool = F.layer_norm(pool, (cfg.cap_feat_dim,), sd["cap_embedder.0.weight"], sd["cap_embedder.0.bias"], 1e-5)
    cap_emb = _linear(_r(pool, bf16), sd["cap_embedder.1.weight"], sd["cap_embedder.1.bias"], bf16)
    adaln_input = _r(te + cap_emb, bf16)
    scale = math.sqrt(math.log(N, base_seqlen) / hd) if proportional_attn else math.sqrt(1 / hd)  # model.py:374-377
    L = cfg.n_layers if n_layers is None else n_layers
    hidden = []
    for i in range(L):
        h = flag_block(sd, i, cfg, h, freqs_cis, cf, cap_mask, adaln_input, scale, bf16)
        if return_hidden:
            hidden.append(h)
    o = _final_shift_scale(sd, cfg, h, adaln_input, bf16)
    oc = cfg.out_channels
    img = o.view(B, Hp, Wp + 1, p, p, oc)[:, :, :-1]  # unpatchify, return_tensor path (:750-757)
    img = img.permute(0, 5, 1, 3, 2, 4).flatten(4, 5).flatten(2, 3)
    if cfg.learn_sigma:
        img = img.chunk(2, dim=1)[0]
    return (img, hidden) if return_hidden else img


def flag_forward_with_cfg(sd, cfg: NextDiTConfig, x, t, cap_feats, cap_mask, cfg_scale, rope_scaling_factor=None,
                          ntk_factor=None, base_seqlen=None, proportional_attn=False, bf16=False, n_layers=None):
    """DiT_Llama.forward_with_cfg (model.py:866-922); None factors keep the constructor's table (1.0, 1.0)."""
    p = cfg.patch_size
    N = (x.shape[2] // p) * (x.shape[3] // p + 1)
    table = rope_table_1d(cfg.head_dim, N, rope_scaling_factor=1.0 if rope_scaling_factor is None else rope_scaling_factor,
                          ntk_factor=1.0 if ntk_factor is None else ntk_factor)
    half = x[: len(x) // 2]
    out = flag_forward(sd, cfg, torch.cat([half, half], dim=0), t, cap_feats, cap_mask, freqs_table=table,
                       proportional_attn=proportional_attn, base_seqlen=base_seqlen, bf16=bf16, n_layers=n_layers)
    return _cfg_combine(out, cfg_scale, bf16)
