"""Thin PyTorch-side plumbing around the C-ABI engine: tensors in, tensors out, current HIP stream.

PyTorch only supplies device memory, streams and (elsewhere) ``torch.distributed``; every FLOP of the
denoising path runs in the HIP kernels behind ``liblumina_dit.so``.
"""
from __future__ import annotations

import ctypes as C
import math
from dataclasses import dataclass
from typing import Dict, Iterable, Optional, Tuple

import torch

from . import _lib
from ._lib import LtConfig, LtStepArgs, LuminaLibError

_DT = {torch.float32: _lib.LT_F32, torch.bfloat16: _lib.LT_BF16, torch.float16: _lib.LT_F16}


def _stream_ptr(device: torch.device) -> int:
    return torch.cuda.current_stream(device).cuda_stream


def _require_gpu(t: torch.Tensor, name: str) -> None:
    if not t.is_cuda:
        raise LuminaLibError(
            f"{name} lives on {t.device}; the MI355X denoising engine only runs on a ROCm device "
            "(there is no CPU fallback - use oracle/ for CPU reference numbers in tests)"
        )


class _SourceCache:
    """Remembers WHICH tensors the engine's conditioning was prepared from, so that the per-step calls of an ODE solve (same
    ``model_kwargs`` objects every step) do not redo the caption work.  A hit needs the very same tensor objects, unmodified
    (``_version``); the cache keeps them referenced, so their storage cannot be freed and handed to a different prompt that
    would then look identical by address."""

    def __init__(self):
        self._src = None
        self._key = None

    @staticmethod
    def _describe(tensors, extra):
        return (tuple((t._version, tuple(t.shape), t.dtype, t.device) for t in tensors), tuple(extra))

    @staticmethod
    def _trackable(tensors) -> bool:
        # inference tensors (torch.inference_mode) carry no version counter: in-place edits cannot be seen, so never cache them
        return not any(t.is_inference() for t in tensors)

    def hit(self, tensors, extra=()) -> bool:
        return (self._src is not None and self._trackable(tensors) and len(self._src) == len(tensors)
                and all(a is b for a, b in zip(self._src, tensors)) and self._key == self._describe(tensors, extra))

    def store(self, tensors, extra=()) -> None:
        if not self._trackable(tensors):
            self.clear()
            return
        self._src, self._key = tuple(tensors), self._describe(tensors, extra)

    def clear(self) -> None:
        self._src = self._key = None


@dataclass
class EngineLimits:
    max_batch: int = 2
    max_tokens: int = 4096
    max_text: int = 256


class DiTEngine:
    """Owns one ``lt_engine`` handle (weights arena + workspace in HBM) for one model on one GPU."""

    def __init__(self, *, variant: int, dim: int, n_layers: int, n_heads: int, n_kv_heads: int, ffn_hidden: int,
                 patch_size: int, in_channels: int, out_channels: int, cap_feat_dim: int, qk_norm: bool,
                 norm_eps: float, num_classes: int = 0, num_experts: int = 0, limits: Optional[EngineLimits] = None,
                 device: Optional[torch.device] = None):
        self.lib = _lib.load()
        self.device = torch.device(device if device is not None else "cuda")
        if not torch.cuda.is_available():
            raise LuminaLibError("no ROCm device visible: the denoising engine cannot run (no CPU fallback)")
        lim = limits or EngineLimits()
        self.limits = lim
        cfg = LtConfig(
            variant=variant, dim=dim, n_layers=n_layers, n_heads=n_heads, n_kv_heads=n_kv_heads,
            ffn_hidden=ffn_hidden, patch_size=patch_size, in_channels=in_channels, out_channels=out_channels,
            cap_feat_dim=cap_feat_dim, adaln_dim=min(dim, 1024), qk_norm=int(bool(qk_norm)), num_classes=num_classes,
            norm_eps=norm_eps, max_batch=lim.max_batch, max_tokens=lim.max_tokens, max_text=lim.max_text,
            rope_table_len=384, num_experts=num_experts,
        )
        self.cfg = cfg
        self.in_channels = in_channels
        self.patch_size = patch_size
        handle = C.c_void_p()
        with torch.cuda.device(self.device):
            _lib.check(self.lib.lt_create(C.byref(cfg), C.byref(handle)), "lt_create")
        self.handle = handle
        self._prompt = _SourceCache()
        self.num_classes = num_classes

    def __del__(self):
        h = getattr(self, "handle", None)
        if h:
            try:
                self.lib.lt_destroy(h)
            except Exception:
                pass
            self.handle = None

    # ---- options (include/lumina_dit_debug.h) ----------------------------------------------------------
    def set_option(self, name: str, value: Optional[int]) -> None:
        """override one kernel-selection option for THIS engine only (``None`` drops the override: the process default set by
        ``lt_set_option`` applies again)"""
        v = _lib.LT_OPTION_INHERIT if value is None else int(value)
        _lib.check(self.lib.lt_engine_set_option(self.handle, name.encode(), v), f"lt_engine_set_option({name})")
        self._prompt.clear()  # the hoisted text K / V may depend on the option (rmsnorm_apex): prepare the conditioning again

    def get_option(self, name: str) -> int:
        """the value in effect for this engine (its override, else the process default)"""
        out = C.c_int32(0)
        _lib.check(self.lib.lt_engine_get_option(self.handle, name.encode(), C.byref(out)), f"lt_engine_get_option({name})")
        return int(out.value)

    # ---- weights ---------------------------------------------------------------------------------
    def load_state_dict(self, state: Dict[str, torch.Tensor], skip: Iterable[str] = ()) -> None:
        """Upload every tensor of a reference-format state_dict into the engine."""
        # the hoisted conditioning (text K / V of every layer, caption / label embedding) was computed from the OLD weights:
        # forget which tensors it came from, so the next call prepares it again (lt_set_weight also invalidates it engine-side)
        self._prompt.clear()
        s = _stream_ptr(self.device)
        skip = set(skip)
        with torch.cuda.device(self.device):
            for key, t in state.items():
                if key in skip:
                    continue
                t = t.detach()
                if t.device != self.device or t.dtype not in _DT or not t.is_contiguous():
                    dt = t.dtype if t.dtype in _DT else torch.float32
                    t = t.to(device=self.device, dtype=dt).contiguous()
                shape = (C.c_int64 * max(t.dim(), 1))(*(t.shape if t.dim() else (1,)))
                rc = self.lib.lt_set_weight(self.handle, key.encode(), C.c_void_p(t.data_ptr()), _DT[t.dtype], shape,
                                            max(t.dim(), 1), C.c_void_p(s))
                _lib.check(rc, f"lt_set_weight({key})")
            torch.cuda.current_stream(self.device).synchronize()  # sources may be temporaries
            _lib.check(self.lib.lt_weights_ready(self.handle), "lt_weights_ready")

    # ---- prompt ----------------------------------------------------------------------------------
    def prepare_prompt(self, cap_feats: torch.Tensor, cap_mask: torch.Tensor) -> None:
        _require_gpu(cap_feats, "cap_feats")
        if self._prompt.hit((cap_feats, cap_mask), ("prompt",)):
            return
        feats = cap_feats if cap_feats.dtype in (torch.float32, torch.bfloat16) else cap_feats.float()
        feats = feats.contiguous()
        mask = cap_mask.to(device=feats.device, dtype=torch.int32).contiguous()
        B, T, _ = feats.shape
        with torch.cuda.device(self.device):
            rc = self.lib.lt_prepare_prompt(self.handle, C.c_void_p(feats.data_ptr()), _DT[feats.dtype],
                                            C.c_void_p(mask.data_ptr()), B, T, C.c_void_p(_stream_ptr(self.device)))
        _lib.check(rc, "lt_prepare_prompt")
        self._keep = (feats, mask)  # keep the temporaries alive until the stream has consumed them
        self._prompt.store((cap_feats, cap_mask), ("prompt",))

    def prepare_prompt_regional(self, cap_feats: torch.Tensor, cap_mask: torch.Tensor, global_feats: torch.Tensor,
                                global_mask: torch.Tensor, h_split: int, w_split: int) -> None:
        """compositional Next-DiT: Y captions (regions of the cond row ..., uncond row) + the one-row global caption"""
        _require_gpu(cap_feats, "cap_feats")
        src, extra = (cap_feats, cap_mask, global_feats, global_mask), ("regional", int(h_split), int(w_split))
        if self._prompt.hit(src, extra):
            return
        dt = cap_feats.dtype if cap_feats.dtype in (torch.float32, torch.bfloat16) else torch.float32
        feats = cap_feats.to(dt).contiguous()
        gfeats = global_feats.to(device=feats.device, dtype=dt).contiguous()
        mask = cap_mask.to(device=feats.device, dtype=torch.int32).contiguous()
        gmask = global_mask.to(device=feats.device, dtype=torch.int32).contiguous()
        Y, T, _ = feats.shape
        with torch.cuda.device(self.device):
            rc = self.lib.lt_prepare_prompt_regional(self.handle, C.c_void_p(feats.data_ptr()), _DT[feats.dtype],
                                                     C.c_void_p(mask.data_ptr()), Y, T, C.c_void_p(gfeats.data_ptr()),
                                                     C.c_void_p(gmask.data_ptr()), gfeats.shape[1], int(h_split), int(w_split),
                                                     C.c_void_p(_stream_ptr(self.device)))
        _lib.check(rc, "lt_prepare_prompt_regional")
        self._keep = (feats, mask, gfeats, gmask)
        self._prompt.store(src, extra)

    def prepare_labels(self, y: torch.Tensor) -> None:
        """class-conditional variants: y int [B] (null class = num_classes, Next-DiT-ImageNet/sample.py:181)"""
        _require_gpu(y, "y")
        if self._prompt.hit((y,), ("labels",)):
            return
        lab = y.to(dtype=torch.int32).contiguous()
        with torch.cuda.device(self.device):
            rc = self.lib.lt_prepare_labels(self.handle, C.c_void_p(lab.data_ptr()), lab.numel(), C.c_void_p(_stream_ptr(self.device)))
        _lib.check(rc, "lt_prepare_labels")
        self._keep = (lab,)
        self._prompt.store((y,), ("labels",))

    # ---- one model evaluation --------------------------------------------------------------------
    def _step_args(self, x: torch.Tensor, cfg_scale: float, scale_factor: float, scale_watershed: float,
                   base_seqlen: Optional[int], proportional_attn: bool, cfg_channels: int = 3,
                   ntk_factor: float = 1.0) -> LtStepArgs:
        B, Cc, H, W = x.shape
        if x.dtype not in (torch.float32, torch.bfloat16):
            raise LuminaLibError(f"state dtype {x.dtype} unsupported (bf16 or fp32)")
        return LtStepArgs(cfg_scale=float(cfg_scale), scale_factor=float(scale_factor),
                          scale_watershed=float(scale_watershed), base_seqlen=int(base_seqlen or 0),
                          proportional_attn=int(bool(proportional_attn)), latent_h=H, latent_w=W, batch=B,
                          io_dtype=_DT[x.dtype], cfg_channels=cfg_channels, ntk_factor=float(ntk_factor))

    def forward(self, x: torch.Tensor, t: torch.Tensor, *, use_cfg: bool, cfg_scale: float = 1.0,
                scale_factor: float = 1.0, scale_watershed: float = 1.0, base_seqlen: Optional[int] = None,
                proportional_attn: bool = False, ntk_factor: float = 1.0) -> torch.Tensor:
        _require_gpu(x, "x")
        x = x.contiguous()
        t32 = t.to(device=x.device, dtype=torch.float32).contiguous()
        out = torch.empty_like(x)
        a = self._step_args(x, cfg_scale, scale_factor, scale_watershed, base_seqlen, proportional_attn, ntk_factor=ntk_factor)
        fn = self.lib.lt_forward_cfg if use_cfg else self.lib.lt_forward
        with torch.cuda.device(self.device):
            rc = fn(self.handle, C.c_void_p(x.data_ptr()), C.c_void_p(t32.data_ptr()), C.c_void_p(out.data_ptr()),
                    C.byref(a), C.c_void_p(_stream_ptr(self.device)))
        _lib.check(rc, "lt_forward_cfg" if use_cfg else "lt_forward")
        return out

    def forward_packed(self, xs, t: torch.Tensor, *, scale_factor: float = 1.0, scale_watershed: float = 0.0,
                       base_seqlen: Optional[int] = None, proportional_attn: bool = False):
        """NextDiT.forward on a LIST of [C, H_b, W_b] latents (reference model.py:789-834): one padded batch on the engine,
        one output tensor per sample (lt_forward_packed)."""
        xs = [x.contiguous() for x in xs]
        for x in xs:
            _require_gpu(x, "x")
            if x.dim() != 3 or x.dtype != xs[0].dtype or x.device != xs[0].device:
                raise LuminaLibError("packed forward: every sample must be a [C, H, W] tensor of one dtype on one device")
        B = len(xs)
        t32 = t.to(device=xs[0].device, dtype=torch.float32).contiguous()
        outs = [torch.empty((self.in_channels,) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device) for x in xs]
        a = self._step_args(xs[0].unsqueeze(0), 1.0, scale_factor, scale_watershed, base_seqlen, proportional_attn)
        a.batch, a.latent_h, a.latent_w = B, 0, 0
        xp = (C.c_void_p * B)(*[x.data_ptr() for x in xs])
        op = (C.c_void_p * B)(*[o.data_ptr() for o in outs])
        hw = (C.c_int32 * (2 * B))(*[v for x in xs for v in x.shape[1:]])
        with torch.cuda.device(self.device):
            rc = self.lib.lt_forward_packed(self.handle, xp, hw, C.c_void_p(t32.data_ptr()), op, C.byref(a),
                                            C.c_void_p(_stream_ptr(self.device)))
        _lib.check(rc, "lt_forward_packed")
        return outs

    # ---- whole trajectory -------------------------------------------------------------------------
    def sample_ode(self, z: torch.Tensor, tgrid: torch.Tensor, method: str, *, use_cfg: bool, cfg_scale: float = 1.0,
                   scale_factor: float = 1.0, scale_watershed: float = 1.0, base_seqlen: Optional[int] = None,
                   proportional_attn: bool = False, t_round_to_state_dtype: bool = True,
                   return_trajectory: bool = True, ntk_factor: float = 1.0) -> torch.Tensor:
        _require_gpu(z, "z")
        if method not in _lib.ODE_METHODS:
            raise LuminaLibError(f"fixed-grid method '{method}' not in {sorted(_lib.ODE_METHODS)}")
        z = z.contiguous()
        grid = [float(v) for v in tgrid.detach().to("cpu", torch.float32).tolist()]
        n = len(grid)
        garr = (C.c_float * n)(*grid)
        a = self._step_args(z, cfg_scale, scale_factor, scale_watershed, base_seqlen, proportional_attn, ntk_factor=ntk_factor)
        if return_trajectory:
            out = torch.empty((n,) + tuple(z.shape), dtype=z.dtype, device=z.device)
            traj_ptr, fin_ptr = C.c_void_p(out.data_ptr()), C.c_void_p(0)
        else:
            out = torch.empty_like(z)
            traj_ptr, fin_ptr = C.c_void_p(0), C.c_void_p(out.data_ptr())
        with torch.cuda.device(self.device):
            rc = self.lib.lt_sample_ode(self.handle, C.c_void_p(z.data_ptr()), traj_ptr, fin_ptr, garr, n,
                                        _lib.ODE_METHODS[method], int(use_cfg), int(t_round_to_state_dtype), C.byref(a),
                                        C.c_void_p(_stream_ptr(self.device)))
        _lib.check(rc, "lt_sample_ode")
        return out

    def last_nfe(self) -> int:
        return int(self.lib.lt_last_nfe(self.handle))

    def graph_replays(self) -> int:
        """model evaluations served by a captured HIP graph so far"""
        return int(self.lib.lt_graph_replays(self.handle))

    # ---- mixture-of-experts routing hooks (parity tests) ----------------------------------------------
    def moe_routing_record(self, on: bool = True) -> None:
        _lib.check(self.lib.lt_moe_routing_record(self.handle, 1 if on else 0), "lt_moe_routing_record")

    def moe_routing_read(self, rows: int):
        """experts the last forward picked: int32 array [n_layers, 2 (time, token branch), rows, 2]; -1 = branch not run"""
        import numpy as np
        out = np.empty((int(self.cfg.n_layers), 2, rows, 2), dtype=np.int32)
        _lib.check(self.lib.lt_moe_routing_read(self.handle, out.ctypes.data_as(C.POINTER(C.c_int32)), rows), "lt_moe_routing_read")
        return out

    def moe_routing_force(self, sel=None) -> None:
        """sel int32 [n_layers, 2, rows, 2] (ascending expert ids per row) replaces the top-2 choice of the following forwards;
        None ends it"""
        import numpy as np
        if sel is None:
            _lib.check(self.lib.lt_moe_routing_force(self.handle, None, 0), "lt_moe_routing_force")
            return
        sel = np.ascontiguousarray(sel, dtype=np.int32)
        assert sel.ndim == 4 and sel.shape[0] == int(self.cfg.n_layers) and sel.shape[1] == 2 and sel.shape[3] == 2, sel.shape
        _lib.check(self.lib.lt_moe_routing_force(self.handle, sel.ctypes.data_as(C.POINTER(C.c_int32)), sel.shape[2]),
                   "lt_moe_routing_force")

    # ---- profiling hooks used by bench.py -----------------------------------------------------------
    def profile_enable(self, on) -> None:
        """False / 0: off; True: every kernel class; int: bit mask (1 GEMM, 2 attention, 4 other)"""
        mask = 7 if on is True else int(on)
        _lib.check(self.lib.lt_profile_enable_mask(self.handle, mask), "lt_profile_enable_mask")

    def profile_set_budget(self, klass: int, max_event_launches: int) -> None:
        _lib.check(self.lib.lt_profile_set_budget(self.handle, klass, max_event_launches), "lt_profile_set_budget")

    def profile_set_window(self, klass: int, skip_launches: int, max_event_launches: int) -> None:
        _lib.check(self.lib.lt_profile_set_window(self.handle, klass, skip_launches, max_event_launches), "lt_profile_set_window")

    def profile_reset(self) -> None:
        _lib.check(self.lib.lt_profile_reset(self.handle), "lt_profile_reset")

    def profile_read(self, klass: int) -> Tuple[float, int, float]:
        ms, n, fl = C.c_double(), C.c_int64(), C.c_double()
        _lib.check(self.lib.lt_profile_read(self.handle, klass, C.byref(ms), C.byref(n), C.byref(fl)), "lt_profile_read")
        return ms.value, n.value, fl.value


def ffn_hidden_dim(dim: int, multiple_of: int, ffn_dim_multiplier: Optional[float]) -> int:
    """FeedForward hidden width rule of the reference (lumina_next_t2i/models/model.py:469-473)."""
    hidden = int(2 * (4 * dim) / 3)
    if ffn_dim_multiplier is not None:
        hidden = int(ffn_dim_multiplier * hidden)
    return multiple_of * ((hidden + multiple_of - 1) // multiple_of)


def softmax_scale(seqlen: int, head_dim: int, proportional_attn: bool, base_seqlen: Optional[int]) -> float:
    """model.py:373-376"""
    if proportional_attn:
        return math.sqrt(math.log(seqlen, base_seqlen) / head_dim)
    return math.sqrt(1 / head_dim)
