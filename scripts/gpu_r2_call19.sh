#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for i in 1 2; do for g in 0 1 2; do
python bench.py --no-cpu-baseline --opt gemm_stagger=$g 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
r=d['roofline']
print('gemm_stagger=$g', {k:round(d[k],2) for k in ('value','ms_per_step')}, {k:(round(v,2) if isinstance(v,float) else v) for k,v in d['kernel_time_ms_per_step'].items() if k!='note'}, 'gemm TF/s', round(r['achieved'],1))
"
done; done
