#!/usr/bin/env python
"""Per-dispatch view of a rocprofv3 --kernel-trace CSV: durations of the kernels whose name contains <pattern>, grouped by the
kernel that ran right before them on the queue (the summary CSV only has min / avg / max; a bimodal kernel shows up here as two
predecessor groups).   python scripts/trace_by_predecessor.py <dir or kernel_trace.csv> <pattern> [<pattern> ...]"""
import csv
import glob
import os
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    return re.sub(r"\(.*$", "", name)[:60]


def main():
    src = sys.argv[1]
    files = [src] if os.path.isfile(src) else glob.glob(os.path.join(src, "**", "*kernel_trace.csv"), recursive=True)
    rows = []
    for f in files:
        for r in csv.DictReader(open(f)):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    print(f"{len(rows)} dispatches from {len(files)} file(s)")
    for pat in sys.argv[2:]:
        groups = defaultdict(list)
        gaps = defaultdict(list)
        for i, (s, e, n) in enumerate(rows):
            if pat in n and i > 0:
                groups[short(rows[i - 1][2])].append((e - s) / 1e3)
                gaps[short(rows[i - 1][2])].append((s - rows[i - 1][1]) / 1e3)
        print(f"== {pat}")
        for k, v in sorted(groups.items(), key=lambda kv: -len(kv[1])):
            v2 = sorted(v)
            g = sorted(gaps[k])
            print(f"  after {k:60s} n {len(v):5d}  us min {v2[0]:7.1f} med {v2[len(v2) // 2]:7.1f} max {v2[-1]:7.1f}   gap med {g[len(g) // 2]:6.1f}")


if __name__ == "__main__":
    main()
