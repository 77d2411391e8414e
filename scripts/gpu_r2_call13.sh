#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r2/call13
mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -q -x -k "attention" > $OUT/pytest.log 2>&1; echo "pytest exit $?"; tail -3 $OUT/pytest.log
python scripts/attn_trace_v4.py 2>&1 | grep -v amdgpu.ids | head -12
timeout 300 python scripts/opbench.py attn --rounds 7 --attn-variants 3,4 2>&1 | grep -v amdgpu.ids | tail -3
for v in 3 4 3 4; do
python bench.py --no-cpu-baseline --opt attention_variant=$v 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
r=d['roofline']
print('attention_variant=$v', {k:round(d[k],2) for k in ('value','ms_per_step')}, {k:(round(v,2) if isinstance(v,float) else v) for k,v in d['kernel_time_ms_per_step'].items() if k!='note'}, 'gemm TF/s', round(r['achieved'],1), 'attn', round(d['attention_tflops_per_s'],1))
"
done
