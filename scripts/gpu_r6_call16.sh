#!/bin/bash
# round 6, session 4: attention prologue probe - the first K / V^T tiles' LDS-DMA issued behind the Q rows' global loads (build LT_ATTN_Q_FIRST) against the default order
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r6/call16; mkdir -p $OUT
cd $R
QF=$R/lumina-t2x_amd/lib/qf/liblumina_dit.so
LUMINA_DIT_LIB=$QF timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -q -x -k "qraw or qstat" 2>&1 | tail -2
for i in 1 2 3; do for lib in "" $QF; do
  LUMINA_DIT_LIB=$lib timeout 600 python bench.py --no-cpu-baseline > $OUT/ab.tmp 2>/dev/null; python - $OUT/ab.tmp "lib=${lib:-default}" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2][-30:], round(d["ms_per_step"], 3), [round(x, 3) for x in d["ms_per_step_repeats"]], {k: round(v, 3) for k, v in d["kernel_time_ms_per_step"].items() if isinstance(v, float)}, "attn TF/s", round(d["attention_tflops_per_s"], 1))
PY
done; done | tee $OUT/ab_attn_q_first.log
