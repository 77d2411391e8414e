#!/bin/bash
# bench A/B on ONE box (box-to-box spread is ~6 %): default, graph off, twice each, interleaved
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out/r2/ab
for i in 1 2; do for g in 1 0; do
python bench.py --no-cpu-baseline --opt graph=$g 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
r=d['roofline']
print('graph=$g', {k:round(d[k],2) for k in ('value','ms_per_step','hip_graph_replays')}, {k:(round(v,2) if isinstance(v,float) else v) for k,v in d['kernel_time_ms_per_step'].items() if k!='note'}, 'gemm TF/s', round(r['achieved'],1), 'avg launch ms', round(r['avg_launch_ms'],4), 'attn', round(d['attention_tflops_per_s'],1))
"
done; done
