#!/bin/bash
# The scaling curve the builder could never measure (every lease was ONE GPU): run on an 8-GPU MI355X node.
#
#   bash scripts/scale_sweep.sh [out_file]        (default profiles/scale_curve.jsonl; one bench.py JSON line per run)
#
#  * BASELINE configs[1] (Next-DiT 2B, 1024^2, CFG): bench.py --gpus 1 2 4 8 - weak scaling, one image per GPU, weights replicated, ONE RCCL
#    broadcast of the text features before the timed region, no collective inside it; `value` = images x tokens x NFE / max-over-ranks wall
#  * BASELINE configs[3] in its own form ("batch = 8 sharded over 8 GPUs"): bench.py --workload cfg4 --gpus 8 (and --gpus 1 as its base)
# Each line's `comm` object says what the collective layer saw: backend (nccl = RCCL), rccl_version, ranks_seen = an all-reduce of ones
# (must equal --gpus), every rank's own ms / step and device name.  Efficiency = value(N) / (N x value(1)); expected ~1 (no per-step
# communication, 5 GB of replicated state per GPU) - an expectation until this script has run.
set -u
cd "$(dirname "$0")/.."
OUT=${1:-profiles/scale_curve.jsonl}
export HSA_ENABLE_IPC_MODE_LEGACY=0   # dmabuf IPC only on these hosts (RCCL needs it)
: > "$OUT"
for n in 1 2 4 8; do
  echo "== cfg2 --gpus $n" >&2
  python bench.py --gpus $n --no-cpu-baseline | tail -1 >> "$OUT"
done
for n in 1 8; do
  echo "== cfg4 --gpus $n" >&2
  python bench.py --workload cfg4 --gpus $n --steps 8 --warmup 2 --no-cpu-baseline | tail -1 >> "$OUT"
done
python - "$OUT" <<'PY'
import json, sys
rows = [json.loads(l) for l in open(sys.argv[1]) if l.strip().startswith("{")]
base = {}
for r in rows:
    wl = r["config"]["workload"][:22]
    n = r["n_gpus"]
    base.setdefault(wl, r["value"] / n if n == 1 else None)
    b = base[wl]
    c = r.get("comm", {})
    print(f"{wl:24s} gpus {n}: {r['value']:10.0f} {r['unit']}  {r['ms_per_step']:7.2f} ms/step  efficiency {r['value'] / (n * b):.3f}" if b else
          f"{wl:24s} gpus {n}: {r['value']:10.0f} {r['unit']}", "| backend", c.get("backend"), "rccl", c.get("rccl_version"), "ranks_seen", c.get("ranks_seen"))
PY
