"""Imports the UNMODIFIED reference modules from /root/reference behind the stub packages.
TEST INFRASTRUCTURE ONLY, and only usable in the authoring container (the GPU box has no /root/reference):
it exists to pin ``nextdit_oracle`` against the real reference and to generate ``tests/golden/``.
"""
import importlib
import os
import sys

import torch

REFERENCE_ROOT = os.environ.get("LUMINA_REFERENCE_ROOT", "/root/reference")
_STUBS = os.path.join(os.path.dirname(os.path.abspath(__file__)), "stubs")
_REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def available() -> bool:
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "lumina_next_t2i", "models"))


def load_reference(package: str = "lumina_next_t2i"):
    """Returns (models module, transport module) of the given reference sub-project."""
    if not available():
        raise RuntimeError(f"reference checkout not found under {REFERENCE_ROOT}")
    for p in (_REPO, _STUBS, os.path.join(REFERENCE_ROOT, package)):
        if p not in sys.path:
            sys.path.insert(0, p)
    # model.py:952 hard-codes .cuda(); on the CPU harness make it a no-op
    if not torch.cuda.is_available():
        torch.Tensor.cuda = lambda self, *a, **k: self
    for name in ("models", "transport"):
        if name in sys.modules and not getattr(sys.modules[name], "__file__", "").startswith(REFERENCE_ROOT):
            del sys.modules[name]
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        models = importlib.import_module("models")
        transport = importlib.import_module("transport")
    return models, transport


def build_reference_model(cfg, state_dict):
    """Reference NextDiT (fp32, eval) holding the given weights."""
    models, _ = load_reference()
    model = importlib.import_module("models.model").NextDiT(**cfg.ctor_kwargs()).eval()
    missing = model.load_state_dict(state_dict, strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys
    return model
