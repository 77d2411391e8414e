#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r2/call10
mkdir -p $OUT
cd $R
timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py -m gpu -q -x -k "qk_norm or switches or golden or packed" > $OUT/pytest.log 2>&1; echo "pytest exit $?"; tail -3 $OUT/pytest.log
timeout 200 python scripts/opbench.py elem --rounds 7 2>&1 | tail -8
for i in 1 2; do
python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print({k:round(d[k],2) for k in ('value','ms_per_step')}, {k:(round(v,2) if isinstance(v,float) else v) for k,v in d['kernel_time_ms_per_step'].items() if k!='note'})
"
done
