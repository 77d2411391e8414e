#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r2/call14
mkdir -p $OUT
cd $R
timeout 1500 python -m pytest tests -m gpu -q -x > $OUT/pytest_full.log 2>&1; echo "pytest exit $?"; tail -4 $OUT/pytest_full.log
python bench.py > $OUT/bench.json.log 2>/dev/null; tail -1 $OUT/bench.json.log | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print({k:d[k] for k in ('value','ms_per_step')}, d['kernel_time_ms_per_step'], d['roofline']['achieved'], d['attention_tflops_per_s'])"
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
