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


def copy_root():
    """a scratch directory with the files of the verified archive ``oracle/_ref/reference_files.tar`` (oracle/build_ref.py: byte-identical
    reference files, every sha256 re-checked on extraction), or None when no archive travelled"""
    from oracle import build_ref
    return build_ref.extract() if build_ref.verify() else None


def timing_root():
    """where bench.cpu_baseline() imports the UNMODIFIED reference from: the checkout in the authoring container, the travelling copy
    on the GPU box, None when neither exists (the baseline then falls back to the restatement and says ``kind: "port"``)"""
    return REFERENCE_ROOT if available() else copy_root()


def load_reference(package: str = "lumina_next_t2i", root: str = None):
    """Returns (models module, transport module) of the given reference sub-project."""
    root = root or REFERENCE_ROOT
    if not os.path.isdir(os.path.join(root, package, "models")):
        raise RuntimeError(f"reference sub-project {package} not found under {root}")
    for p in (_REPO, _STUBS, os.path.join(root, package)):
        if p not in sys.path:
            sys.path.insert(0, p)
    # model.py:952 hard-codes .cuda(); on the CPU harness make it a no-op
    if not torch.cuda.is_available():
        torch.Tensor.cuda = lambda self, *a, **k: self
    for name in ("models", "transport"):
        if name in sys.modules and not getattr(sys.modules[name], "__file__", "").startswith(root):
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



def load_anagrams_solvers():
    """The reference's only in-tree statement of an ODE step: ``midpoint_solver`` (``visual_anagrams/generate.py:212-219``), what
    torchdiffeq's fixed-grid ``midpoint`` does per interval (its first half is the Euler slope).  generate.py is a script - importing
    it parses ``sys.argv`` and loads checkpoints at module level (generate.py:268-270) - so the function definitions are compiled
    VERBATIM from the file's syntax tree (no source text is copied or edited) into a namespace that holds ``torch``.
    ``.to("cuda")`` (generate.py:215,218) is made a no-op on the CPU harness exactly as ``.cuda()`` is for model.py:952."""
    import ast
    if not available():
        raise RuntimeError(f"reference checkout not found under {REFERENCE_ROOT}")
    path = os.path.join(REFERENCE_ROOT, "visual_anagrams", "generate.py")
    tree = ast.parse(open(path).read(), filename=path)
    wanted = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in ("midpoint_solver",)]
    assert len(wanted) == 1 and wanted[0].lineno == 212, [(n.name, n.lineno) for n in wanted]
    ns = {"torch": torch}
    exec(compile(ast.Module(body=wanted, type_ignores=[]), path, "exec"), ns)
    if not torch.cuda.is_available() and not getattr(torch.Tensor.to, "_lt_cpu_harness", False):
        _to = torch.Tensor.to

        def to(self, *a, **k):
            a = tuple(x for x in a if not (isinstance(x, str) and x.startswith("cuda")))
            if isinstance(k.get("device"), str) and k["device"].startswith("cuda"):
                k.pop("device")
            return _to(self, *a, **k) if (a or k) else self
        to._lt_cpu_harness = True
        torch.Tensor.to = to
    return ns["midpoint_solver"]
