#!/bin/bash
# round 6, session 3, call 2: LDS-DMA prefetch distance 4 (build parameter LT_W4Q_PD) against 3 at HEAD - headline, three interleaved rounds; the other configs once each
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r6/call6; mkdir -p $OUT
cd $R
PD4=$R/lumina-t2x_amd/lib/pd4/liblumina_dit.so
for i in 1 2 3; do for lib in "" $PD4; do
  LUMINA_DIT_LIB=$lib timeout 600 python bench.py --no-cpu-baseline > $OUT/ab.tmp 2>/dev/null; python - $OUT/ab.tmp "lib=${lib:-default}" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2][-40:], round(d["ms_per_step"], 3), [round(x, 3) for x in d["ms_per_step_repeats"]], {k: round(v, 3) for k, v in d["kernel_time_ms_per_step"].items() if isinstance(v, float)}, "gemm TF/s", round(d["roofline"]["achieved"], 1), "W", round((d.get("power") or {}).get("avg_w") or 0))
PY
done; done | tee $OUT/ab_pd4_headline.log
for lib in "" $PD4; do echo "lib=${lib:-default}"; LUMINA_DIT_LIB=$lib timeout 900 python scripts/bench_configs.py cfg1 cfg3 cfg5 cfg5-1024 --nfe 8 2>&1 | grep "ms/NFE" | cut -c1-60; done | tee $OUT/ab_pd4_configs.log
