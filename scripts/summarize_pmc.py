"""Aggregate rocprofv3 --pmc counter_collection CSVs per kernel name: mean counter value per dispatch.
With --json OUT also writes the HBM-traffic summary bench.py reads for roofline.traffic:
per launch of the dominant kernel class (every gemm_bf16_* dispatch), bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024
(FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE tallies 128-B requests at 64 B, so wide coalesced reads
show exactly half their bytes - MI355X_MICROARCH.md 'HBM')."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

args = sys.argv[1:]
out_json = None
if "--json" in args:
    i = args.index("--json")
    out_json = args[i + 1]
    args = args[:i] + args[i + 2:]

tot = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
for d in args:
    for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
        with open(path) as f:
            for row in csv.DictReader(f):
                k = row.get("Kernel_Name", "?")
                c = row.get("Counter_Name", "?")
                v = float(row.get("Counter_Value", 0) or 0)
                for a in (acc[k[:60]][c], tot[k][c]):
                    a[0] += v
                    a[1] += 1
        print("==", path)
        for k, cs in sorted(acc.items()):
            print(k, {c: (round(a[0] / max(a[1], 1), 1), a[1]) for c, a in cs.items()})

if out_json:
    fetch = [0.0, 0]
    write = [0.0, 0]
    per_kernel = {}
    for k, cs in tot.items():
        if "gemm_bf16" not in k:
            continue
        f, w = cs.get("FETCH_SIZE", [0.0, 0]), cs.get("WRITE_SIZE", [0.0, 0])
        fetch[0] += f[0]; fetch[1] += f[1]
        write[0] += w[0]; write[1] += w[1]
        per_kernel[k[:80]] = {"launches": max(f[1], w[1]), "fetch_kib_avg": f[0] / max(f[1], 1), "write_kib_avg": w[0] / max(w[1], 1)}
    if fetch[1] and write[1]:
        fk, wk = fetch[0] / fetch[1], write[0] / write[1]
        json.dump({"kernel_class": "gemm_bf16_* (all GEMM dispatches of bench.py --steps 2)",
                   "fetch_size_kib_avg": fk, "write_size_kib_avg": wk,
                   "hbm_bytes_per_launch": (2.0 * fk + wk) * 1024.0,
                   "formula": "(2 * FETCH_SIZE + WRITE_SIZE) * 1024; gfx950 FETCH_SIZE counts wide reads at half (MI355X_MICROARCH.md)",
                   "per_kernel": per_kernel}, open(out_json, "w"), indent=1)
        print("wrote", out_json)
