#!/usr/bin/env python3
"""per-kernel register / scratch usage of one HIP source, from hipcc's resource remarks (CPU container, no GPU needed):
    python scripts/kernel_resources.py gemm_bf16.hip [name filter] [-- extra hipcc flags]"""
import os
import re
import subprocess
import sys

csrc = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "lumina-t2x_amd", "csrc")
args = sys.argv[1:]
extra = []
if "--" in args:
    extra = args[args.index("--") + 1:]
    args = args[:args.index("--")]
src, flt = args[0], (args[1] if len(args) > 1 else "")
out = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", *extra, "-c", src, "-o", "/tmp/kr.o",
                      "-Rpass-analysis=kernel-resource-usage"], cwd=csrc, capture_output=True, text=True).stderr
rows, cur = [], None
for line in out.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = {"name": subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()}
        rows.append(cur)
        continue
    m = re.search(r" (VGPRs Spill|SGPRs Spill|VGPRs|AGPRs|ScratchSize|Occupancy)( \[[^\]]*\])?: (\d+)", line)
    if m and cur is not None:
        cur[m.group(1)] = int(m.group(3))
if not rows:
    print(out[-3000:])
for r in rows:
    if flt in r["name"]:
        g = lambda k: r.get(k, 0)
        print("%-100s V %3d A %3d scratch %4d spillV %3d spillS %3d occ %d" % (r["name"][:100], g("VGPRs"), g("AGPRs"), g("ScratchSize"), g("VGPRs Spill"),
                                                                         g("SGPRs Spill"), g("Occupancy")))
