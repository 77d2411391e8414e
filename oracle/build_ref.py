"""Fills the git-ignored ``oracle/_ref/`` with BYTE-IDENTICAL copies of the reference files the CPU-baseline leg imports - TEST /
MEASUREMENT INFRASTRUCTURE ONLY ("the reference itself timed on the bench host in the same run").

    python -m oracle.build_ref            (authoring container; called by __graft_entry__.build() when /root/reference exists)

The reference is pure Python, so "building" it is copying: the files are copied, never edited (MANIFEST.json holds the sha256
of every source next to the sha256 of its copy), they stay out of git history (``oracle/_ref/`` is in .gitignore) and they travel
to the GPU box with the repo snapshot exactly like the built ``.so`` (not in .gpurunignore).  On the GPU box, where /root/reference
does not exist, ``oracle/ref_harness.py`` imports the modules from here behind the same stub packages (oracle/stubs) and
``bench.cpu_baseline()`` times the UNMODIFIED ``NextDiT.forward_with_cfg`` (``kind: "reference"``).  Nothing under
``lumina-t2x_amd/`` may import this directory (tests/test_abi.py).
"""
import hashlib
import json
import os
import shutil
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REF_SRC = os.environ.get("LUMINA_REFERENCE_ROOT", "/root/reference")
REF_DST = os.path.join(HERE, "_ref")

# the sub-projects the harness imports `models` / `transport` from (oracle/ref_harness.py, oracle/make_*_golden.py)
FILES = [
    "lumina_next_t2i/models/__init__.py", "lumina_next_t2i/models/model.py", "lumina_next_t2i/models/components.py",
    "lumina_next_t2i/transport/__init__.py", "lumina_next_t2i/transport/integrators.py", "lumina_next_t2i/transport/path.py",
    "lumina_next_t2i/transport/transport.py", "lumina_next_t2i/transport/utils.py",
]


def _sha(path):
    with open(path, "rb") as f:
        return hashlib.sha256(f.read()).hexdigest()


def build(verbose=True) -> bool:
    """returns True when oracle/_ref holds a complete, verified copy"""
    if not os.path.isdir(os.path.join(REF_SRC, "lumina_next_t2i", "models")):
        ok = verify()
        if verbose:
            print(f"oracle/_ref: no reference checkout under {REF_SRC}; existing copy {'verified' if ok else 'absent'}")
        return ok
    manifest = {}
    for rel in FILES:
        src, dst = os.path.join(REF_SRC, rel), os.path.join(REF_DST, rel)
        os.makedirs(os.path.dirname(dst), exist_ok=True)
        shutil.copyfile(src, dst)
        a, b = _sha(src), _sha(dst)
        assert a == b, rel
        manifest[rel] = a
    with open(os.path.join(REF_DST, "MANIFEST.json"), "w") as f:
        json.dump({"source_root": REF_SRC, "sha256": manifest,
                   "note": "byte-identical copies of reference files for the CPU-baseline timing; git-ignored, never edited"}, f, indent=1)
    if verbose:
        print(f"oracle/_ref: {len(manifest)} reference files copied unmodified (sha256 in MANIFEST.json)")
    return True


def verify() -> bool:
    """the copy is complete and every file still has the sha256 recorded when it was copied from the reference"""
    path = os.path.join(REF_DST, "MANIFEST.json")
    if not os.path.exists(path):
        return False
    with open(path) as f:
        man = json.load(f)["sha256"]
    return all(os.path.exists(os.path.join(REF_DST, rel)) and _sha(os.path.join(REF_DST, rel)) == h for rel, h in man.items()) and \
        set(man) == set(FILES)


if __name__ == "__main__":
    sys.exit(0 if build() else 1)
