"""Engine plumbing shared by the model classes of every family: the modules only HOLD parameters under the
reference's state_dict keys; a forward call hands them to the HIP engine (csrc/) through the C ABI.  There is no
PyTorch implementation of the forward pass and no CPU fallback."""
from __future__ import annotations

from typing import Optional

import torch
import torch.nn as nn

from .. import _lib
from ..engine import DiTEngine, EngineLimits


class EngineBackedModel(nn.Module):
    """Mixin-style base: subclasses set ``_variant`` and implement ``_engine_kwargs()``."""

    _variant: int = _lib.LT_VARIANT_NEXT_T2I

    def _init_engine_state(self) -> None:
        self.engine_limits = EngineLimits()
        self._engine: Optional[DiTEngine] = None
        self._weights_sig = None

    def _signature(self):
        # (inference tensors carry no version counter: in-place edits of such parameters are not seen - rebuild the model then)
        return tuple((p.data_ptr(), -1 if p.is_inference() else p._version) for p in self.parameters())

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
