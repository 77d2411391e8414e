#!/bin/bash
# round 6, call 3: the full GPU suite on the fixed streaming row kernel, the headline profile set, grn_ystat A/B, the bench line
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r6/call3; mkdir -p $OUT
cd $R
timeout 1500 python -m pytest tests -m gpu -q -x -s > $OUT/pytest_gpu_full_suite.log 2>&1; echo "pytest exit $?"; tail -3 $OUT/pytest_gpu_full_suite.log
PROF_OUT=$OUT/prof bash scripts/gpu_prof.sh > $OUT/prof.log 2>&1; grep -E "exit|^\"void (lt_|\(anon)" $OUT/prof.log | head -14 | cut -c1-170
for i in 1 2; do for v in 0 1; do
  timeout 600 python bench.py --no-cpu-baseline --opt grn_ystat=$v > $OUT/ab.tmp 2>/dev/null
  python - $OUT/ab.tmp "grn_ystat=$v" <<'PY' | tee -a $OUT/ab_grn_ystat.log
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], "ms/step", round(d["ms_per_step"], 3), [round(x, 3) for x in d["ms_per_step_repeats"]], {k: round(v, 3) for k, v in d["kernel_time_ms_per_step"].items() if isinstance(v, float)}, "W", round(d["power"]["avg_w"]))
PY
done; done
timeout 900 python bench.py > $OUT/bench_full.json 2> $OUT/bench_full.err; echo "bench exit $?"; cut -c1-300 $OUT/bench_full.json
