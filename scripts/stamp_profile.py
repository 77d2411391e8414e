#!/usr/bin/env python3
"""Adds the commit a counter summary was taken at to a pmc_*.json on its way into profiles/ (the GPU box has no .git; the summary already
carries the sha256 of its kernel sources, which is what bench.py checks):   python scripts/stamp_profile.py profiles/r06/pmc_gemm.json ..."""
import json
import subprocess
import sys

head = subprocess.run(["git", "rev-parse", "HEAD"], capture_output=True, text=True).stdout.strip()
dirty = subprocess.run(["git", "status", "--porcelain", "--", "lumina-t2x_amd/csrc"], capture_output=True, text=True).stdout.strip()
for path in sys.argv[1:]:
    d = json.load(open(path))
    d["git_head"] = head + (" + uncommitted changes under csrc/ (the sha256 stamps are the sources profiled)" if dirty else "")
    json.dump(d, open(path, "w"), indent=1)
    print(path, "<-", d["git_head"])
