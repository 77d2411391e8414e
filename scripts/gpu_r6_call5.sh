#!/bin/bash
# round 6, session 3, call 1: attention text-tile skip + hd-96 tail split - parity tests, then same-box A/Bs (cfg 3 for the split, headline for the skip)
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r6/call5; mkdir -p $OUT
cd $R
timeout 1500 python -m pytest tests/test_gpu_ops.py -m gpu -q -x -k "attention" > $OUT/pytest_attention.log 2>&1; echo "pytest exit $?"; tail -5 $OUT/pytest_attention.log
for i in 1 2; do for v in 0 4; do echo "attn_tail_split=$v"; timeout 600 python scripts/bench_configs.py cfg3 --nfe 8 --opt attn_tail_split=$v 2>&1 | grep -E "ms/NFE" | tail -2; done; done | tee $OUT/ab_tail_split_cfg3.log
for i in 1 2; do for v in 0 1; do timeout 600 python bench.py --no-cpu-baseline --opt attn_text_skip=$v > $OUT/ab.tmp 2>/dev/null; python - $OUT/ab.tmp "attn_text_skip=$v" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], round(d["ms_per_step"], 3), [round(x, 3) for x in d["ms_per_step_repeats"]], {k: round(v, 3) for k, v in d["kernel_time_ms_per_step"].items() if isinstance(v, float)}, "attn TF/s", round(d["attention_tflops_per_s"], 1), "W", round((d.get("power") or {}).get("avg_w") or 0))
PY
done; done | tee $OUT/ab_text_skip.log
