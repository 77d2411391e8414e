#!/bin/bash
# rocprofv3 kernel trace of the headline bench, reduced on the box to per-predecessor duration groups (scripts/trace_by_predecessor.py)
set -u
cd /tmp && export TMPDIR=/tmp
export LT_NO_EVENT_PROFILE=1
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r3/trace_pred; mkdir -p $OUT
cd $R
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_tp -o bench -- python bench.py --steps 6 --warmup 2 --no-cpu-baseline > $OUT/trace.log 2>&1
echo "trace exit $?"
python scripts/trace_by_predecessor.py /tmp/prof_tp gated_residual_norm qk_norm_rope_pair attn_fwd_kernel_v4 "gemm_bf16_w4q<0" "gemm_bf16_w4q<1" "gemm_bf16_w4q<3" 2>&1 | tee $OUT/by_predecessor.txt | cut -c1-200
