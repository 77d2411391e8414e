"""Engine plumbing shared by the model classes of every family: the modules only HOLD parameters under the
reference's state_dict keys; a forward call hands them to the HIP engine (csrc/) through the C ABI.  There is no
PyTorch implementation of the forward pass and no CPU fallback."""
from __future__ import annotations

from typing import Optional

import torch
import torch.nn as nn

from .. import _lib
from ..engine import DiTEngine, EngineLimits


# Any (re-)registration of a Parameter OR of a sub-module anywhere (setattr of an nn.Parameter, load_state_dict(assign=True),
# `m.layers[0] = prebuilt_block`, ...) bumps this epoch; the cached flat MODULE lists below are rebuilt when it moves.  In-place edits
# and .to() / .half() conversions keep the Parameter objects and show up as a changed (data_ptr, version) pair instead; parameters
# that disappear without a registration (`del m.a.bias`, `m.a.bias = None`) show up because the signature is read from the live
# `_parameters` dicts of the cached modules, not from a cached parameter list (the round-3 form cached the parameters and
# was driven by the parameter hook alone - a pre-built sub-module swapped in, or a parameter removed, kept the engine on OLD weights).
_PARAM_EPOCH = [0]


def _on_parameter_registration(module, name, param):
    _PARAM_EPOCH[0] += 1
    return None


def _on_module_registration(module, name, submodule):
    _PARAM_EPOCH[0] += 1
    return None


torch.nn.modules.module.register_module_parameter_registration_hook(_on_parameter_registration)
torch.nn.modules.module.register_module_module_registration_hook(_on_module_registration)


class WeightWatch:
    """'did any weight change since the engine last loaded them?' at ~50 us per call instead of ~0.3 ms: the module tree (567
    parameters under ~230 modules for the 2B model) is walked once and the flat module list kept; every per-step call of a
    host-driven sampler (dopri5, SDE: the default of Next-DiT-ImageNet/sample.py) goes through this check."""

    def _watch_reset(self) -> None:
        self._watch_modules = None
        self._watch_epoch = -1

    def _apply(self, fn, *args, **kwargs):  # .to() / .cuda() / .half(): may swap the Parameter objects without registering them
        out = super()._apply(fn, *args, **kwargs)
        _PARAM_EPOCH[0] += 1
        return out

    def load_state_dict(self, *args, **kwargs):
        out = super().load_state_dict(*args, **kwargs)
        _PARAM_EPOCH[0] += 1
        return out

    def _signature(self):
        # REMOVING a sub-module fires no registration hook (ModuleList.pop / __delitem__, `del m.sub`, edits of `_modules`): the number of
        # children over the cached tree is the structural term that catches it - a removal shrinks its parent's `_modules`
        stale = getattr(self, "_watch_modules", None) is None or self._watch_epoch != _PARAM_EPOCH[0]
        if not stale and sum(len(m._modules) for m in self._watch_tree) != self._watch_children:
            stale = True
        if stale:
            self._watch_tree = list(self.modules())
            self._watch_children = sum(len(m._modules) for m in self._watch_tree)
            self._watch_modules = [m for m in self._watch_tree if m._parameters]
            self._watch_epoch = _PARAM_EPOCH[0]
        # (inference tensors carry no version counter: in-place edits of such parameters are not seen - rebuild the model then)
        return tuple((p.data_ptr(), -1 if p.is_inference() else p._version)
                     for m in self._watch_modules for p in m._parameters.values() if p is not None)


class EngineBackedModel(WeightWatch, nn.Module):
    """Mixin-style base: subclasses set ``_variant`` and implement ``_engine_kwargs()``."""

    _variant: int = _lib.LT_VARIANT_NEXT_T2I

    def _init_engine_state(self) -> None:
        self.engine_limits = EngineLimits()
        self._engine: Optional[DiTEngine] = None
        self._weights_sig = None
        self._watch_reset()

    def _engine_kwargs(self) -> dict:
        raise NotImplementedError

    def _tokens(self, H: int, W: int) -> int:
        return (H // self.patch_size) * (W // self.patch_size)

    def engine(self, x: torch.Tensor, text_len: int = 0) -> DiTEngine:
        """Create / resize the engine for this call's shapes and make sure it holds the current weights."""
        if not x.is_cuda:
            raise _lib.LuminaLibError(
                f"{type(self).__name__}.forward needs tensors on a ROCm device: the MI355X engine has no CPU fallback")
        B, _, H, W = x.shape
        lim = self.engine_limits
        need = EngineLimits(max(lim.max_batch, B), max(lim.max_tokens, self._tokens(H, W)), max(lim.max_text, text_len))
        if self._engine is None or need != self._engine.limits or self._engine.device != x.device:
            self._engine = None
            self._engine = DiTEngine(variant=self._variant, limits=need, device=x.device, **self._engine_kwargs())
            self.engine_limits = need
            self._weights_sig = None
        sig = self._signature()
        if sig != self._weights_sig:
            self._engine.load_state_dict(self.state_dict())
            self._weights_sig = sig
        return self._engine

    def parameter_count(self) -> int:
        return sum(p.numel() for p in self.parameters())

    def get_fsdp_wrap_module_list(self):
        return list(self.layers)

    def get_checkpointing_wrap_module_list(self):
        return list(self.layers)
