"""Fills the git-ignored ``oracle/_ref/`` with ONE archive of byte-identical reference files for the CPU-baseline leg - TEST / MEASUREMENT
INFRASTRUCTURE ONLY ("the reference itself timed on the bench host in the same run").

    python -m oracle.build_ref            (authoring container; called by __graft_entry__.build() when /root/reference exists)

The reference is pure Python, so "building" it is packing: the files are read where they lie under /root/reference and written, unedited, into
``oracle/_ref/reference_files.tar`` with a MANIFEST.json beside it (sha256 of every member and of the archive).  No reference source file ever
sits in the repository tree: the archive is a build artefact like the ``.so`` - ``oracle/_ref/`` is in .gitignore (out of history) and not in
.gpurunignore (it travels to the GPU box with the snapshot).  On the GPU box, where /root/reference does not exist,
``oracle/ref_harness.timing_root()`` extracts the archive into a scratch directory, re-checks every sha256 and imports the modules from there
behind the same stub packages (oracle/stubs); ``bench.cpu_baseline()`` then times the UNMODIFIED ``NextDiT.forward_with_cfg``
(``kind: "reference"``).  Nothing under ``lumina-t2x_amd/`` may import this directory (tests/test_abi.py).
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


ARCHIVE = os.path.join(REF_DST, "reference_files.tar")
MANIFEST = os.path.join(REF_DST, "MANIFEST.json")


def _sha_bytes(data):
    return hashlib.sha256(data).hexdigest()


def _sha(path):
    with open(path, "rb") as f:
        return _sha_bytes(f.read())


def build(verbose=True) -> bool:
    """returns True when oracle/_ref holds a complete, verified archive"""
    import io
    import tarfile
    if not os.path.isdir(os.path.join(REF_SRC, "lumina_next_t2i", "models")):
        ok = verify()
        if verbose:
            print(f"oracle/_ref: no reference checkout under {REF_SRC}; existing archive {'verified' if ok else 'absent'}")
        return ok
    os.makedirs(REF_DST, exist_ok=True)
    for stale in ("lumina_next_t2i",):  # (the first form of this script kept loose copies)
        shutil.rmtree(os.path.join(REF_DST, stale), ignore_errors=True)
    manifest = {}
    with tarfile.open(ARCHIVE, "w") as tar:
        for rel in FILES:
            with open(os.path.join(REF_SRC, rel), "rb") as f:
                data = f.read()
            manifest[rel] = _sha_bytes(data)
            info = tarfile.TarInfo(rel)
            info.size = len(data)
            info.mtime = 0
            tar.addfile(info, io.BytesIO(data))
    with open(MANIFEST, "w") as f:
        json.dump({"source_root": REF_SRC, "sha256": manifest, "archive_sha256": _sha(ARCHIVE),
                   "note": "byte-identical reference files for the CPU-baseline timing, packed unedited; git-ignored build artefact"}, f, indent=1)
    if verbose:
        print(f"oracle/_ref: {len(manifest)} reference files packed unmodified into reference_files.tar (sha256 in MANIFEST.json)")
    return verify()


def verify() -> bool:
    """the archive is the one the manifest describes (cheap: one hash); members are re-checked on extraction"""
    if not (os.path.exists(MANIFEST) and os.path.exists(ARCHIVE)):
        return False
    with open(MANIFEST) as f:
        man = json.load(f)
    return set(man["sha256"]) == set(FILES) and _sha(ARCHIVE) == man.get("archive_sha256")


_extracted = [None]


def extract() -> str:
    """unpacks the verified archive into a scratch directory (once per process), checks every member against the manifest, returns the root"""
    import tarfile
    import tempfile
    if _extracted[0] and os.path.isdir(_extracted[0]):
        return _extracted[0]
    assert verify(), "oracle/_ref holds no verified archive"
    with open(MANIFEST) as f:
        man = json.load(f)["sha256"]
    root = tempfile.mkdtemp(prefix="lumina_ref_")
    with tarfile.open(ARCHIVE, "r") as tar:
        for m in tar.getmembers():
            assert m.name in man and m.isfile() and not m.name.startswith(("/", "..")), m.name
            data = tar.extractfile(m).read()
            assert _sha_bytes(data) == man[m.name], m.name
            dst = os.path.join(root, m.name)
            os.makedirs(os.path.dirname(dst), exist_ok=True)
            with open(dst, "wb") as f:
                f.write(data)
    _extracted[0] = root
    return root


if __name__ == "__main__":
    sys.exit(0 if build() else 1)
