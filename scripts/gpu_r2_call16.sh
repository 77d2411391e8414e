#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r2/call16
mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py -m gpu -q -x -k "fused_qkv or switches or vt_epilogue or golden" > $OUT/pytest.log 2>&1; echo "pytest exit $?"; tail -5 $OUT/pytest.log
for v in 1 0 1 0; do
python bench.py --no-cpu-baseline --opt qkv_fused_gemm=$v 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
r=d['roofline']
print('qkv_fused_gemm=$v', {k:round(d[k],2) for k in ('value','ms_per_step')}, {k:(round(v,2) if isinstance(v,float) else v) for k,v in d['kernel_time_ms_per_step'].items() if k!='note'}, 'gemm TF/s', round(r['achieved'],1), 'attn', round(d['attention_tflops_per_s'],1))
"
done
