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
#   bash scripts/scale_sweep.sh --dry-run         prints the exact commands (the driver's launch form for N > 1) and runs nothing
set -u
cd "$(dirname "$0")/.."
DRY=0
if [ "${1:-}" = "--dry-run" ]; then DRY=1; shift; fi
OUT=${1:-profiles/scale_curve.jsonl}
export HSA_ENABLE_IPC_MODE_LEGACY=0   # dmabuf IPC only on these hosts (RCCL needs it)
PORT=${MASTER_PORT:-29531}
# N = 1: plain python; N > 1: one rank per GPU through torch.distributed.run on 127.0.0.1, exactly as the round driver launches bench.py
# (bench.py reads RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment; `python bench.py --gpus N` alone would self-launch the same way)
cmd() {  # cmd <gpus> <extra bench args...>
  local n=$1; shift
  if [ "$n" = 1 ]; then echo "python bench.py --gpus 1 $* --no-cpu-baseline"
  else echo "python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $PORT bench.py --gpus $n $* --no-cpu-baseline"; fi
}
CMDS=()
for n in 1 2 4 8; do CMDS+=("$(cmd $n --steps 29 --warmup 3)"); done                                  # BASELINE configs[1], 1 / 2 / 4 / 8 GPUs
for n in 1 2 4 8; do CMDS+=("$(cmd $n --workload cfg4 --steps 8 --warmup 2)"); done                    # BASELINE configs[3]: an image per GPU, up to "8 over 8"
if [ $DRY = 1 ]; then
  echo "# HSA_ENABLE_IPC_MODE_LEGACY=0 exported; one JSON line per command appended to $OUT"
  for c in "${CMDS[@]}"; do echo "$c"; done
  exit 0
fi
: > "$OUT"
for c in "${CMDS[@]}"; do
  echo "== $c" >&2
  $c | tail -1 >> "$OUT"
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
