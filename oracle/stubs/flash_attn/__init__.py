"""Stand-in for ``flash_attn`` (not installed anywhere offline).  TEST INFRASTRUCTURE ONLY.

The fp32 CPU path of the reference never reaches these (model.py:378 takes the SDPA branch).  The bf16 yardstick runs of
oracle/make_fulldepth_golden.py (``refbf16_*``: the UNMODIFIED reference module moved to bfloat16) do: there the two entry points the
reference calls are provided as what the published kernel computes up to its tiling - softmax(scale * q k^T) v with fp32 scores,
fp32 accumulation, one cast of the output back to the input dtype - on top of torch's CPU SDPA (its flash path: no N x N tensor).
``LUMINA_FLASH_STUB=pbf16`` additionally rounds the un-normalised probabilities to the input dtype before the P V product, as the
published kernel (and the engine's MFMA operand) does; the default is the plain fp32 form the round-5 verdict asked for.
Like the package it stands in for, half precision only: an fp32 call raises (the fp32 reference must not call it).
"""
import os

import torch
import torch.nn.functional as F


def _check(q, k, v, dropout_p, causal):
    if q.dtype not in (torch.bfloat16, torch.float16):
        raise RuntimeError("flash_attn stand-in: half-precision inputs only (the fp32 reference path must not call it)")
    if dropout_p != 0.0 or causal:
        raise RuntimeError("flash_attn stand-in: the sampling path uses dropout_p = 0, causal = False")
    assert q.dtype == k.dtype == v.dtype


def _sdpa(q, k, v, scale):
    """q [H, Nq, hd], k / v [Hkv, Nk, hd] in half precision -> [H, Nq, hd] in the same dtype"""
    H, Hkv = q.shape[0], k.shape[0]
    dt = q.dtype
    if scale is None:
        scale = q.shape[-1] ** -0.5
    if H != Hkv:
        k = k.repeat_interleave(H // Hkv, dim=0)
        v = v.repeat_interleave(H // Hkv, dim=0)
    if os.environ.get("LUMINA_FLASH_STUB", "sdpa") == "pbf16":
        out = torch.empty(q.shape, dtype=dt)
        kf, vf = k.float(), v
        for h in range(H):
            for s in range(0, q.shape[1], 2048):
                sc = (q[h, s:s + 2048].float() @ kf[h].T) * scale
                p = torch.exp(sc - sc.amax(dim=-1, keepdim=True))
                l = p.sum(dim=-1, keepdim=True)
                o = p.to(dt).float() @ vf[h].float()
                out[h, s:s + 2048] = (o / l).to(dt)
        return out
    return F.scaled_dot_product_attention(q.float()[None], k.float()[None], v.float()[None], scale=scale)[0].to(dt)


def flash_attn_func(q, k, v, dropout_p=0.0, softmax_scale=None, causal=False, **_):
    """q [B, Nq, H, hd], k / v [B, Nk, Hkv, hd] -> [B, Nq, H, hd]"""
    _check(q, k, v, dropout_p, causal)
    return torch.stack([_sdpa(q[b].transpose(0, 1), k[b].transpose(0, 1), v[b].transpose(0, 1), softmax_scale).transpose(0, 1)
                        for b in range(q.shape[0])])


def flash_attn_varlen_func(q, k, v, cu_seqlens_q, cu_seqlens_k, max_seqlen_q, max_seqlen_k, dropout_p=0.0, softmax_scale=None,
                           causal=False, **_):
    """packed q [sum Nq, H, hd], k / v [sum Nk, Hkv, hd]; sequence b is rows cu_seqlens[b] : cu_seqlens[b + 1]"""
    _check(q, k, v, dropout_p, causal)
    out = torch.empty(q.shape, dtype=q.dtype)
    cq, ck = cu_seqlens_q.tolist(), cu_seqlens_k.tolist()
    for b in range(len(cq) - 1):
        qs, ks = slice(cq[b], cq[b + 1]), slice(ck[b], ck[b + 1])
        out[qs] = _sdpa(q[qs].transpose(0, 1), k[ks].transpose(0, 1), v[ks].transpose(0, 1), softmax_scale).transpose(0, 1)
    return out
