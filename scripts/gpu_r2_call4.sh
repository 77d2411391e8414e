#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r2/call4
mkdir -p $OUT
cd $R
timeout 600 python -m pytest tests/test_gpu_model.py tests/test_gpu_samplers.py -m gpu -q -x > $OUT/pytest.log 2>&1; echo "pytest exit $?"; tail -8 $OUT/pytest.log
timeout 600 python -m pytest tests/test_gpu_variants.py -m gpu -q -x -k "moe or driver" > $OUT/pytest2.log 2>&1; echo "pytest2 exit $?"; tail -5 $OUT/pytest2.log
for gopt in 1 0; do
timeout 200 python bench.py --no-cpu-baseline --opt graph=$gopt > $OUT/bench_graph$gopt.json 2> $OUT/bench.err; echo "bench graph=$gopt exit $?"; python - <<PY
import json
d=json.loads(open("$OUT/bench_graph$gopt.json").read().strip().splitlines()[-1])
print({k:d[k] for k in ("value","ms_per_step","hip_graph_replays")}, d["kernel_time_ms_per_step"], d["roofline"]["achieved"])
PY
done
timeout 300 python scripts/bench_configs.py > $OUT/bench_configs.log 2>&1; echo "bench_configs exit $?"; tail -12 $OUT/bench_configs.log
