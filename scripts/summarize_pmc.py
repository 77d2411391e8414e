"""Aggregate rocprofv3 --pmc counter_collection CSVs per kernel name: mean counter value per dispatch.
With --json OUT also writes the HBM-traffic summary bench.py reads for roofline.traffic:
per launch of the dominant kernel class (every gemm_bf16_* dispatch), bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024
(FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE tallies 128-B requests at 64 B, so wide coalesced reads
show exactly half their bytes - MI355X_MICROARCH.md 'HBM')."""
import csv
import glob
import hashlib
import json
import os
import sys
from collections import defaultdict

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(REPO, "lumina-t2x_amd", "csrc")


def source_stamp(names):
    """sha256 of the kernel sources a counter summary belongs to: bench.py refuses a committed summary whose sources have changed since
    (VERDICT r5 item 6); `git_head` is added by scripts/stamp_profile.py when the file is copied into profiles/ (the GPU box has no .git)"""
    return {n: hashlib.sha256(open(os.path.join(CSRC, n), "rb").read()).hexdigest() for n in names}


args = sys.argv[1:]
out_json = out_attn = None
if "--json" in args:
    i = args.index("--json")
    out_json = args[i + 1]
    args = args[:i] + args[i + 2:]
if "--json-attn" in args:
    i = args.index("--json-attn")
    out_attn = args[i + 1]
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
                   "source_sha256": source_stamp(["gemm_device.h", "gemm_bf16.hip"]),
                   "per_kernel": per_kernel}, open(out_json, "w"), indent=1)
        print("wrote", out_json)

if out_attn:
    # the hd-72 self-attention kernel of the headline step (bench.py `roofline_attention`): fabric traffic and matrix-pipe duty per launch
    for k, cs in tot.items():
        if "attn_fwd_kernel_v4<72>" not in k:
            continue
        avg = {c: a[0] / max(a[1], 1) for c, a in cs.items()}
        if "FETCH_SIZE" not in avg or "WRITE_SIZE" not in avg:
            continue
        d = {"kernel": k[:80], "launches": max(a[1] for a in cs.values()),
             "fetch_size_kib_avg": avg["FETCH_SIZE"], "write_size_kib_avg": avg["WRITE_SIZE"],
             "hbm_bytes_per_launch": (2.0 * avg["FETCH_SIZE"] + avg["WRITE_SIZE"]) * 1024.0,
             "formula": "(2 * FETCH_SIZE + WRITE_SIZE) * 1024; gfx950 FETCH_SIZE counts wide reads at half (MI355X_MICROARCH.md)",
             "source_sha256": source_stamp(["attention_v4.hip", "attention_v4_asm.inc"])}
        if "SQ_VALU_MFMA_BUSY_CYCLES" in avg and "GRBM_GUI_ACTIVE" in avg and avg["GRBM_GUI_ACTIVE"] > 0:
            # MFMA-busy cycles are summed over the chip's 1024 SIMDs, GRBM_GUI_ACTIVE over its 8 XCDs (PMC passes serialise the kernels)
            d["mfma_busy_cycles_avg"] = avg["SQ_VALU_MFMA_BUSY_CYCLES"]
            d["grbm_gui_active_avg"] = avg["GRBM_GUI_ACTIVE"]
            d["mfma_duty"] = (avg["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0) / (avg["GRBM_GUI_ACTIVE"] / 8.0)
        json.dump(d, open(out_attn, "w"), indent=1)
        print("wrote", out_attn)
