"""MI355X-native Next-DiT / Flag-DiT denoising engine (host-side mirror of the reference Python API).

Layout mirrors a reference sub-project directory (``lumina_next_t2i/``): ``models/``, ``transport/`` and
``sample.py`` keep the reference's names and signatures; the compute lives in ``csrc/`` (HIP kernels +
C ABI) and is reached through ``_lib`` / ``engine``.  The directory name contains a hyphen, so the package
is imported as ``lumina_t2x_amd`` through the one-line shim at the repository root.
"""
from . import _lib  # noqa: F401
from . import models, transport  # noqa: F401
from .engine import DiTEngine, EngineLimits  # noqa: F401

__all__ = ["models", "transport", "DiTEngine", "EngineLimits"]
