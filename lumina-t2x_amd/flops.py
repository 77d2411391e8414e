"""ALGORITHMIC work of one denoising step (one forward of a cond+uncond pair), SURVEY.md 8(d):  2 x MAC of the GEMMs and
of the attention contractions, nothing else.  Used by bench.py for the roofline line; depends on the model's dimensions
only (no weights, no device).  tests/test_host_logic.py holds it to the oracle's independent count."""
from __future__ import annotations


def ffn_hidden(dim: int, multiple_of: int = 256, ffn_dim_multiplier=None) -> int:
    """FeedForward width (lumina_next_t2i/models/model.py:469-473)."""
    h = int(2 * (4 * dim) / 3)
    if ffn_dim_multiplier is not None:
        h = int(ffn_dim_multiplier * h)
    return multiple_of * ((h + multiple_of - 1) // multiple_of)


def flops_per_nfe(*, dim: int, n_layers: int, n_heads: int, n_kv_heads=None, ffn: int = 0, cap_feat_dim: int = 0,
                  n_tokens: int, text_len: int = 0, batch: int = 2, adaln_chunks: int = 4, patch_size: int = 2, in_channels: int = 4,
                  out_channels: int = 8, ffn_visits: int = 1) -> float:
    """per token and layer: q, o (2 d d) + k, v (2 d d_kv) projections, w1/w2/w3 (3 d F), QK^T + PV over N keys (2 N d) and
    over T text keys (2 T d); per text token and layer: wk_y, wv_y (2 cap d_kv); per sample and layer: the adaLN GEMV;
    per token: x_embedder + final linear.  All x 2 (multiply + add).  ``ffn_visits``: FFNs a token runs through per block -
    1 for the dense families, 2 for one top-2 mixture-of-experts FFN (Next-DiT-MoE/models/models.py, models1.py), 4 for the
    time + space pair of models2.py (every token visits two experts in each; SURVEY.md 8d row 5: 1.41 TFLOP at 256 tokens)."""
    hd = dim // n_heads
    dkv = (n_heads if n_kv_heads is None else n_kv_heads) * hd
    F = ffn or ffn_hidden(dim)
    per_tok_layer = 2 * (2 * dim * dim + 2 * dim * dkv) + 6 * dim * F * ffn_visits + 4 * n_tokens * dim + 4 * text_len * dim
    per_txt_layer = 4 * cap_feat_dim * dkv
    per_sample_layer = 2 * min(dim, 1024) * adaln_chunks * dim
    pp = patch_size * patch_size
    embed = 2 * pp * in_channels * dim + 2 * dim * pp * out_channels
    return float(batch) * (n_layers * (n_tokens * per_tok_layer + text_len * per_txt_layer + per_sample_layer) + n_tokens * embed)
