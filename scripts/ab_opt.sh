#!/bin/bash
# interleaved same-box A/B of one option on the headline bench, N rounds: scripts/ab_opt.sh <opt> "<v0> <v1> ..." [rounds] [log]
set -u
OPT=$1; VALS=$2; ROUNDS=${3:-2}; LOG=${4:-/dev/null}
for i in $(seq $ROUNDS); do for v in $VALS; do
  timeout 600 python bench.py --no-cpu-baseline --opt $OPT=$v > /tmp/ab.tmp 2>/dev/null
  python - /tmp/ab.tmp "$OPT=$v" <<'PY' | tee -a $LOG
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], "ms/step", round(d["ms_per_step"], 3), [round(x, 3) for x in d["ms_per_step_repeats"]], {k: round(v, 3) for k, v in d["kernel_time_ms_per_step"].items() if isinstance(v, float)}, "W", round(d["power"]["avg_w"]))
PY
done; done
