#!/bin/bash
# round 6, session 3, call 3: TIMING PROBE - the GEMM's LDS-DMA stream issuing whole 128-byte lines (row-pair-interleaved operand layout,
# addresses only: build LT_W4Q_PAIRLINE, garbage results) against the product's half lines, op level and inside the step
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r6/call7; mkdir -p $OUT
cd $R
PL=$R/lumina-t2x_amd/lib/pl/liblumina_dit.so
for i in 1 2; do for lib in "" $PL; do echo "lib=${lib:-default}"; LUMINA_DIT_LIB=$lib timeout 600 python scripts/opbench.py gemm --gemm-variants 0 --rounds 7 2>&1 | grep -E "^gemm" | cut -c1-120; done; done | tee $OUT/opbench_pairline.log
timeout 600 python scripts/opbench.py gemm_vendor --rounds 7 2>&1 | grep vendor-cmp | cut -c1-140 | tee $OUT/opbench_vendor.log
for i in 1 2; do for lib in "" $PL; do
  LUMINA_DIT_LIB=$lib timeout 600 python bench.py --no-cpu-baseline > $OUT/ab.tmp 2>$OUT/ab.err; python - $OUT/ab.tmp "lib=${lib:-default}" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[2][-40:], round(d["ms_per_step"], 3), [round(x, 3) for x in d["ms_per_step_repeats"]], {k: round(v, 3) for k, v in d["kernel_time_ms_per_step"].items() if isinstance(v, float)}, "gemm TF/s", round(d["roofline"]["achieved"], 1), "W", round((d.get("power") or {}).get("avg_w") or 0))
except Exception as e:
    print(sys.argv[2], "failed", e)
PY
done; done | tee $OUT/ab_pairline_step.log
tail -3 $OUT/ab.err
