#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r2/call7
mkdir -p $OUT
cd $R
timeout 1500 python -m pytest tests -m gpu -q -x -s > $OUT/pytest_gpu.log 2>&1; echo "pytest exit $?"; tail -6 $OUT/pytest_gpu.log; grep "engine vs reference" $OUT/pytest_gpu.log
timeout 300 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench exit $?"; python - <<PY
import json
d=json.loads(open("$OUT/bench.json").read().strip().splitlines()[-1])
print({k:d[k] for k in ("value","ms_per_step","hip_graph_replays")}, d["kernel_time_ms_per_step"], d["roofline"]["achieved"], d["attention_tflops_per_s"])
print(d["roofline"]["kernel"])
PY
timeout 400 python scripts/bench_configs.py cfg3 cfg4 > $OUT/bench_cfg34.log 2>&1; tail -2 $OUT/bench_cfg34.log
