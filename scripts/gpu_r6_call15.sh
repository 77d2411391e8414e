#!/bin/bash
# round 6, session 4: smoke() at HEAD, the shared-device 2-rank bench line, and one A/B (row kernel with two rows per wave on the pair layout)
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r6/call15; mkdir -p $OUT
cd $R
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.log 2>&1; echo "smoke exit $?"; tail -3 $OUT/smoke.log
timeout 900 python bench.py --gpus 2 --share-device --steps 6 --warmup 2 --no-cpu-baseline > $OUT/bench_share2.json 2> $OUT/bench_share2.err; echo "share2 exit $?"; cut -c1-250 $OUT/bench_share2.json; tail -2 $OUT/bench_share2.err
for i in 1 2; do for v in 1 2; do timeout 600 python bench.py --no-cpu-baseline --opt grn_ystat=$v > $OUT/ab.tmp 2>/dev/null; python - $OUT/ab.tmp "grn_ystat=$v" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], round(d["ms_per_step"], 3), [round(x, 3) for x in d["ms_per_step_repeats"]], {k: round(v, 3) for k, v in d["kernel_time_ms_per_step"].items() if isinstance(v, float)})
PY
done; done | tee $OUT/ab_grn_rows_per_wave_pair.log
