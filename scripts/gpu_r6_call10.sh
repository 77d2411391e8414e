#!/bin/bash
# round 6, session 3, call 6: prefetch distance 4 once more, now on top of the pair layout (half the requests in flight per slab)
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r6/call10; mkdir -p $OUT
cd $R
PD4=$R/lumina-t2x_amd/lib/pd4/liblumina_dit.so
for i in 1 2 3; do for lib in "" $PD4; do
  LUMINA_DIT_LIB=$lib timeout 600 python bench.py --no-cpu-baseline > $OUT/ab.tmp 2>/dev/null; python - $OUT/ab.tmp "lib=${lib:-default}" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2][-40:], round(d["ms_per_step"], 3), [round(x, 3) for x in d["ms_per_step_repeats"]], {k: round(v, 3) for k, v in d["kernel_time_ms_per_step"].items() if isinstance(v, float)}, "gemm TF/s", round(d["roofline"]["achieved"], 1), "W", round((d.get("power") or {}).get("avg_w") or 0))
PY
done; done | tee $OUT/ab_pd4_pair.log
