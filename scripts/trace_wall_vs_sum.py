#!/usr/bin/env python
"""Sum of kernel durations against wall time per NFE, from ONE rocprofv3 --kernel-trace CSV (DESIGN.md 9 asserts
"wall = sum of kernel durations" for the 512-row configs - this is the measurement).  An NFE ends with the launch whose name contains
<marker> (default: ode_combine, the Euler update); the first two NFEs (warm-up, graph capture) are dropped.

    python scripts/trace_wall_vs_sum.py <dir or kernel_trace.csv> [marker]

Per NFE: launches, wall = end of its last kernel - end of the previous NFE's last kernel, sum of kernel durations, sum of the idle gaps
between consecutive kernels (start - previous end, negative values = overlap counted as 0), and the five largest gaps with the pair of
kernels around them."""
import csv
import glob
import os
import re
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    return re.sub(r"\(.*$", "", name)[:48]


def main():
    src = sys.argv[1]
    marker = sys.argv[2] if len(sys.argv) > 2 else "ode_combine"
    files = [src] if os.path.isfile(src) else glob.glob(os.path.join(src, "**", "*kernel_trace.csv"), recursive=True)
    rows = []
    for f in files:
        for r in csv.DictReader(open(f)):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    ends = [i for i, r in enumerate(rows) if marker in r[2]]
    print(f"{len(rows)} dispatches, {len(ends)} NFE markers ('{marker}')")
    stats = []
    for a, b in zip(ends[1:-1], ends[2:]):  # NFE = launches a+1 .. b
        seg = rows[a + 1:b + 1]
        wall = (seg[-1][1] - rows[a][1]) / 1e3
        ksum = sum(e - s for s, e, _ in seg) / 1e3
        gaps = []
        prev = rows[a]
        for r in seg:
            gaps.append((max(0, r[0] - prev[1]) / 1e3, short(prev[2]), short(r[2])))
            prev = r
        stats.append((len(seg), wall, ksum, sum(g[0] for g in gaps), sorted(gaps, reverse=True)[:5]))
    if not stats:
        print("no complete NFE found")
        return
    print(f"{'NFE':>4} {'launches':>8} {'wall us':>10} {'sum kernels us':>15} {'sum gaps us':>12} {'kernels / wall':>15}")
    for i, (n, w, k, g, _) in enumerate(stats):
        print(f"{i:4d} {n:8d} {w:10.1f} {k:15.1f} {g:12.1f} {k / w:15.3f}")
    n = len(stats)
    w, k, g = (sum(s[j] for s in stats) / n for j in (1, 2, 3))
    print(f"mean over {n} NFE: wall {w:.1f} us, sum of kernel durations {k:.1f} us ({k / w * 100:.1f} % of wall), idle gaps {g:.1f} us "
          f"({g / w * 100:.1f} %), {stats[-1][0]} launches -> {g / stats[-1][0]:.2f} us of gap per launch")
    print("largest gaps of the last NFE (us | after | before):")
    for gap, p, q in stats[-1][4]:
        print(f"  {gap:7.2f} | {p} | {q}")


if __name__ == "__main__":
    main()
