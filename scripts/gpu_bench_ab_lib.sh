#!/bin/bash
# bench A/B of two builds of the library on ONE box (box-to-box spread is ~6 %): lib/liblumina_dit_old.so vs the default, interleaved
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
N=${1:-3}
for i in $(seq $N); do for g in old new; do
if [ $g = old ]; then export LUMINA_DIT_LIB=$R/lumina-t2x_amd/lib/liblumina_dit_old.so; else unset LUMINA_DIT_LIB; fi
python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
r=d['roofline']
print('$g', {k:round(d[k],2) for k in ('value','ms_per_step')}, {k:(round(v,2) if isinstance(v,float) else v) for k,v in d['kernel_time_ms_per_step'].items() if k!='note'}, 'gemm TF/s', round(r['achieved'],1), 'attn', round(d['attention_tflops_per_s'],1))
"
done; done
