#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r6/call4; mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests -m gpu -q -x -s -k "proj_gated or gated_residual or routing_pinned or trajectory_with" > $OUT/pytest_gpu.log 2>&1; echo "pytest exit $?"; tail -3 $OUT/pytest_gpu.log; grep "routing pinned to" $OUT/pytest_gpu.log | cut -c1-600
for i in 1 2 3; do for v in 1 2; do
  timeout 600 python bench.py --no-cpu-baseline --opt grn_ystat=$v > $OUT/ab.tmp 2>/dev/null
  python - $OUT/ab.tmp "grn_ystat=$v" <<'PY' | tee -a $OUT/ab_grn_rows_per_wave.log
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], "ms/step", round(d["ms_per_step"], 3), [round(x, 3) for x in d["ms_per_step_repeats"]], {k: round(v, 3) for k, v in d["kernel_time_ms_per_step"].items() if isinstance(v, float)}, "W", round(d["power"]["avg_w"]))
PY
done; done
