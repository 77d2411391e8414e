#!/bin/bash
# round 6, session 3, call 4: the row-pair-interleaved operand layout (option pair_layout) - parity tests, then the same-box A/B on the headline
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r6/call8; mkdir -p $OUT
cd $R
timeout 1500 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py -m gpu -q -x -k "pair or path_switches or full_width" > $OUT/pytest_pair.log 2>&1; echo "pytest exit $?"; tail -15 $OUT/pytest_pair.log
for i in 1 2; do for v in 0 1; do timeout 600 python bench.py --no-cpu-baseline --opt pair_layout=$v > $OUT/ab.tmp 2>$OUT/ab.err; python - $OUT/ab.tmp "pair_layout=$v" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[2], round(d["ms_per_step"], 3), [round(x, 3) for x in d["ms_per_step_repeats"]], {k: round(v, 3) for k, v in d["kernel_time_ms_per_step"].items() if isinstance(v, float)}, "gemm TF/s", round(d["roofline"]["achieved"], 1), "W", round((d.get("power") or {}).get("avg_w") or 0))
except Exception as e:
    print(sys.argv[2], "failed", e)
PY
done; done | tee $OUT/ab_pair_layout.log
tail -3 $OUT/ab.err
