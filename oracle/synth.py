"""Deterministic synthetic weights / inputs shared by the oracle, the golden generator, the GPU tests and
bench.py.  numpy's PCG64 stream is bit-reproducible for a fixed numpy version, so fixtures
only need to store seeds, not weights.  TEST INFRASTRUCTURE (see oracle/__init__.py)."""
from __future__ import annotations

from dataclasses import asdict, dataclass
from typing import Dict, Optional

import numpy as np
import torch


@dataclass
class NextDiTConfig:
    dim: int = 2304
    n_layers: int = 24
    n_heads: int = 32
    n_kv_heads: Optional[int] = None
    multiple_of: int = 256
    ffn_dim_multiplier: Optional[float] = None
    norm_eps: float = 1e-5
    qk_norm: bool = True
    cap_feat_dim: int = 2048
    patch_size: int = 2
    in_channels: int = 4
    learn_sigma: bool = True
    # which reference class the config describes:
    #   "next_t2i"  lumina_next_t2i/models/model.py NextDiT            (text-conditional Next-DiT)
    #   "imagenet"  Next-DiT-ImageNet/models/models.py DiT_Llama       (class-conditional Next-DiT, BASELINE configs[0])
    #   "flag_t2i"  lumina_t2i/models/model.py DiT_Llama               (Flag-DiT, BASELINE configs[2])
    #   "moe"       Next-DiT-MoE/models/models2.py DiT_Llama           (time + space MoE, BASELINE configs[4])
    #   "moe_time"  Next-DiT-MoE/models/models.py  DiT_Llama           (one time-routed MoE FFN per block; 8 experts)
    #   "moe_space" Next-DiT-MoE/models/models1.py DiT_Llama           (one token-routed MoE FFN per block; 8 experts)
    family: str = "next_t2i"
    num_classes: int = 1000
    num_experts: int = 4
    num_experts_per_tok: int = 2

    @property
    def chunks(self) -> int:
        return {"next_t2i": 4, "imagenet": 4, "flag_t2i": 6, "moe": 6, "moe_time": 4, "moe_space": 4}[self.family]

    @property
    def has_text(self) -> bool:
        return self.family in ("next_t2i", "flag_t2i")

    @property
    def kv_heads(self) -> int:
        return self.n_heads if self.n_kv_heads is None else self.n_kv_heads

    @property
    def head_dim(self) -> int:
        return self.dim // self.n_heads

    @property
    def ffn_hidden(self) -> int:
        h = int(2 * (4 * self.dim) / 3)  # reference model.py:469-473
        if self.ffn_dim_multiplier is not None:
            h = int(self.ffn_dim_multiplier * h)
        return self.multiple_of * ((h + self.multiple_of - 1) // self.multiple_of)

    @property
    def out_channels(self) -> int:
        return self.in_channels * 2 if self.learn_sigma else self.in_channels

    def ctor_kwargs(self) -> dict:
        kw = dict(patch_size=self.patch_size, in_channels=self.in_channels, dim=self.dim, n_layers=self.n_layers,
                  n_heads=self.n_heads, n_kv_heads=self.n_kv_heads, multiple_of=self.multiple_of,
                  ffn_dim_multiplier=self.ffn_dim_multiplier, norm_eps=self.norm_eps, learn_sigma=self.learn_sigma,
                  qk_norm=self.qk_norm)
        if self.has_text:
            kw["cap_feat_dim"] = self.cap_feat_dim
        else:
            kw["num_classes"] = self.num_classes
        return kw

    def to_dict(self) -> dict:
        return asdict(self)


TINY = NextDiTConfig(dim=576, n_layers=2, n_heads=8, cap_feat_dim=128)            # hd 72, MHA
TINY_GQA = NextDiTConfig(dim=576, n_layers=2, n_heads=8, n_kv_heads=2, cap_feat_dim=128)
NEXT_2B = NextDiTConfig()                                                          # BASELINE cfg 2
TINY_IMAGENET = NextDiTConfig(dim=384, n_layers=2, n_heads=8, family="imagenet", num_classes=10)   # hd 48
IMAGENET_600M = NextDiTConfig(dim=1536, n_layers=16, n_heads=32, family="imagenet")               # BASELINE cfg 1
TINY_FLAG = NextDiTConfig(dim=768, n_layers=2, n_heads=8, cap_feat_dim=128, family="flag_t2i")    # hd 96
FLAG_5B = NextDiTConfig(dim=3072, n_layers=32, n_heads=32, cap_feat_dim=4096, family="flag_t2i")  # BASELINE cfg 3
TINY_MOE = NextDiTConfig(dim=384, n_layers=2, n_heads=8, family="moe", num_classes=10)            # hd 48
TINY_MOE_TIME = NextDiTConfig(dim=384, n_layers=2, n_heads=8, family="moe_time", num_classes=10, num_experts=8)
TINY_MOE_SPACE = NextDiTConfig(dim=384, n_layers=2, n_heads=8, family="moe_space", num_classes=10, num_experts=8)


def _state_shapes_next_t2i(cfg: NextDiTConfig) -> Dict[str, tuple]:
    """SURVEY.md A.2 (order frozen: tests/golden/nextdit_*.npz store seeds, not weights)."""
    d, A, cap, F = cfg.dim, min(cfg.dim, 1024), cfg.cap_feat_dim, cfg.ffn_hidden
    dkv = cfg.kv_heads * cfg.head_dim
    pp = cfg.patch_size * cfg.patch_size
    s: Dict[str, tuple] = {
        "pad_token": (d,), "x_embedder.weight": (d, pp * cfg.in_channels), "x_embedder.bias": (d,),
        "t_embedder.mlp.0.weight": (A, 256), "t_embedder.mlp.0.bias": (A,),
        "t_embedder.mlp.2.weight": (A, A), "t_embedder.mlp.2.bias": (A,),
        "cap_embedder.0.weight": (cap,), "cap_embedder.0.bias": (cap,),
        "cap_embedder.1.weight": (A, cap), "cap_embedder.1.bias": (A,),
        "final_layer.linear.weight": (pp * cfg.out_channels, d), "final_layer.linear.bias": (pp * cfg.out_channels,),
        "final_layer.adaLN_modulation.1.weight": (d, A), "final_layer.adaLN_modulation.1.bias": (d,),
    }
    for i in range(cfg.n_layers):
        p = f"layers.{i}."
        s[p + "attention.gate"] = (cfg.n_heads,)
        s[p + "attention.wq.weight"] = (d, d)
        s[p + "attention.wk.weight"] = (dkv, d)
        s[p + "attention.wv.weight"] = (dkv, d)
        s[p + "attention.wo.weight"] = (d, d)
        s[p + "attention.wk_y.weight"] = (dkv, cap)
        s[p + "attention.wv_y.weight"] = (dkv, cap)
        if cfg.qk_norm:
            for nm, w in (("q_norm", d), ("k_norm", dkv), ("ky_norm", dkv)):
                s[p + f"attention.{nm}.weight"] = (w,)
                s[p + f"attention.{nm}.bias"] = (w,)
        s[p + "feed_forward.w1.weight"] = (F, d)
        s[p + "feed_forward.w2.weight"] = (d, F)
        s[p + "feed_forward.w3.weight"] = (F, d)
        for nm in ("attention_norm1", "attention_norm2", "ffn_norm1", "ffn_norm2"):
            s[p + nm + ".weight"] = (d,)
        s[p + "attention_y_norm.weight"] = (cap,)
        s[p + "adaLN_modulation.1.weight"] = (4 * d, A)
        s[p + "adaLN_modulation.1.bias"] = (4 * d,)
    return s


def state_shapes(cfg: NextDiTConfig) -> Dict[str, tuple]:
    """key -> shape of the reference state_dict (SURVEY.md A.2 for next_t2i; the other families follow their
    reference constructors: Next-DiT-ImageNet/models/models.py:841-895, lumina_t2i/models/model.py:666-741,
    Next-DiT-MoE/models/models2.py:693-768)."""
    if cfg.family == "next_t2i":
        return _state_shapes_next_t2i(cfg)  # key ORDER is part of the fixtures: weights are drawn from one RNG stream
    d, A, cap, F = cfg.dim, min(cfg.dim, 1024), cfg.cap_feat_dim, cfg.ffn_hidden
    dkv = cfg.kv_heads * cfg.head_dim
    pp = cfg.patch_size * cfg.patch_size
    fam = cfg.family
    final_mod = d if fam == "next_t2i" else 2 * d  # scale only vs (shift, scale)
    s: Dict[str, tuple] = {
        "x_embedder.weight": (d, pp * cfg.in_channels), "x_embedder.bias": (d,),
        "t_embedder.mlp.0.weight": (A, 256), "t_embedder.mlp.0.bias": (A,),
        "t_embedder.mlp.2.weight": (A, A), "t_embedder.mlp.2.bias": (A,),
        "final_layer.linear.weight": (pp * cfg.out_channels, d), "final_layer.linear.bias": (pp * cfg.out_channels,),
        "final_layer.adaLN_modulation.1.weight": (final_mod, A), "final_layer.adaLN_modulation.1.bias": (final_mod,),
    }
    if cfg.has_text:
        s.update({"pad_token": (d,), "cap_embedder.0.weight": (cap,), "cap_embedder.0.bias": (cap,),
                  "cap_embedder.1.weight": (A, cap), "cap_embedder.1.bias": (A,)})
        if fam == "flag_t2i":
            s["eol_token"] = (d,)
    else:
        s["y_embedder.embedding_table.weight"] = (cfg.num_classes + 1, A)
    for i in range(cfg.n_layers):
        p = f"layers.{i}."
        s[p + "attention.wq.weight"] = (d, d)
        s[p + "attention.wk.weight"] = (dkv, d)
        s[p + "attention.wv.weight"] = (dkv, d)
        s[p + "attention.wo.weight"] = (d, d)
        norms = [("q_norm", d), ("k_norm", dkv)]
        if cfg.has_text:
            s[p + "attention.gate"] = (cfg.n_heads,)
            s[p + "attention.wk_y.weight"] = (dkv, cap)
            s[p + "attention.wv_y.weight"] = (dkv, cap)
            s[p + "attention_y_norm.weight"] = (cap,)
            norms.append(("ky_norm", dkv))
        if cfg.qk_norm:
            for nm, w in norms:
                s[p + f"attention.{nm}.weight"] = (w,)
                s[p + f"attention.{nm}.bias"] = (w,)
        if fam in ("moe_time", "moe_space"):
            s[p + "feed_forward.gate.weight"] = (cfg.num_experts, A if fam == "moe_time" else d)
            for e in range(cfg.num_experts):
                s[p + f"feed_forward.experts.{e}.w1.weight"] = (F, d)
                s[p + f"feed_forward.experts.{e}.w2.weight"] = (d, F)
                s[p + f"feed_forward.experts.{e}.w3.weight"] = (F, d)
        elif fam == "moe":
            for br, gate_in in (("feed_forward_time", A), ("feed_forward_space", d)):
                s[p + br + ".gate.weight"] = (cfg.num_experts, gate_in)
                for e in range(cfg.num_experts):
                    s[p + f"{br}.experts.{e}.w1.weight"] = (F, d)
                    s[p + f"{br}.experts.{e}.w2.weight"] = (d, F)
                    s[p + f"{br}.experts.{e}.w3.weight"] = (F, d)
        else:
            s[p + "feed_forward.w1.weight"] = (F, d)
            s[p + "feed_forward.w2.weight"] = (d, F)
            s[p + "feed_forward.w3.weight"] = (F, d)
        norm_names = {"next_t2i": ("attention_norm1", "attention_norm2", "ffn_norm1", "ffn_norm2"),
                      "imagenet": ("attention_norm", "ffn_norm"),   # the *_norm1 pre-norms are weight-free (PFRMSNorm)
                      "flag_t2i": ("attention_norm", "ffn_norm"),
                      "moe": ("attention_norm", "ffn_norm_time", "ffn_norm_space"),
                      "moe_time": ("attention_norm", "ffn_norm"), "moe_space": ("attention_norm", "ffn_norm")}[fam]
        for nm in norm_names:
            s[p + nm + ".weight"] = (d,)
        s[p + "adaLN_modulation.1.weight"] = (cfg.chunks * d, A)
        s[p + "adaLN_modulation.1.bias"] = (cfg.chunks * d,)
    return s


def _draw(rng, key: str, shape) -> np.ndarray:
    if key.endswith("attention.gate"):
        return rng.standard_normal(shape, dtype=np.float32) * 0.5
    if key.endswith(".gate.weight"):  # MoE routers: spread the logits so top-2 choices differ per token / sample
        return rng.standard_normal(shape, dtype=np.float32) * (4.0 / np.sqrt(shape[-1]))
    if len(shape) == 1 and key.endswith(".weight"):  # every 1-D weight is a norm weight
        return 1.0 + rng.standard_normal(shape, dtype=np.float32) * 0.02
    if len(shape) == 1:  # biases, pad_token
        return rng.standard_normal(shape, dtype=np.float32) * 0.02
    return rng.standard_normal(shape, dtype=np.float32) * min(0.06, 1.0 / np.sqrt(shape[-1]))


def synth_state_dict(cfg: NextDiTConfig, seed: int = 0, dtype=torch.float32, streams: bool = False) -> Dict[str, torch.Tensor]:
    """SURVEY.md 8d recipe: matrices ~ N(0, s^2) with s = min(0.02 * 3, 1/sqrt(fan_in)) so activations stay O(1)
    through deep stacks, norm weights 1 + N(0, 0.02^2), biases N(0, 0.02^2), attention.gate ~ N(0, 0.5^2).
    (The reference zero-initialises adaLN / final / cap-embedder / gate - model.py:567,643,652,709,201 - which
    would make every block an identity and parity vacuous.)
    streams=False: ONE PCG64 stream consumed in key order (what the seed-only fixtures of tests/golden/*_tiny.npz rely on).
    streams=True: one child stream per key (SeedSequence(seed).spawn), drawn by a thread pool - the multi-billion-parameter
    draws of the full-depth fixtures (tests/golden/full_*.npz) take minutes on one core; same values on any core count."""
    shapes = state_shapes(cfg)
    out: Dict[str, torch.Tensor] = {}
    if not streams:
        rng = np.random.default_rng(seed)
        for key, shape in shapes.items():
            out[key] = torch.from_numpy(np.ascontiguousarray(_draw(rng, key, shape))).to(dtype)
        return out
    import concurrent.futures as cf
    import os
    keys = list(shapes)
    children = np.random.SeedSequence(seed).spawn(len(keys))

    def one(i):
        a = _draw(np.random.default_rng(children[i]), keys[i], shapes[keys[i]])
        return torch.from_numpy(np.ascontiguousarray(a)).to(dtype)

    with cf.ThreadPoolExecutor(max(1, min(48, os.cpu_count() or 1))) as ex:  # numpy releases the GIL while it draws
        for key, t in zip(keys, ex.map(one, range(len(keys)))):
            out[key] = t
    return out


def synth_inputs(cfg: NextDiTConfig, latent_hw=(16, 16), text_len: int = 16, uncond_len: int = 8, seed: int = 1,
                 t_value: float = 0.5):
    """z duplicated for cond+uncond (utils/cli.py:222-223), cap_feats ~ N(0,1), cond mask all ones, uncond mask
    [1]*uncond_len + [0]*rest (empty prompt = BOS + padding to x8, sample.py:38)."""
    rng = np.random.default_rng(seed)
    H, W = latent_hw
    z = torch.from_numpy(rng.standard_normal((1, cfg.in_channels, H, W), dtype=np.float32)).repeat(2, 1, 1, 1)
    t = torch.full((2,), float(t_value), dtype=torch.float32)
    if not cfg.has_text:  # class-conditional: (label, null class) like Next-DiT-ImageNet/sample.py:181
        y = torch.tensor([int(rng.integers(0, cfg.num_classes)), cfg.num_classes], dtype=torch.int64)
        return z, t, y
    cap = torch.from_numpy(rng.standard_normal((2, text_len, cfg.cap_feat_dim), dtype=np.float32))
    mask = torch.ones(2, text_len, dtype=torch.int32)
    mask[1, uncond_len:] = 0
    return z, t, cap, mask
