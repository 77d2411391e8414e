#!/bin/bash
# round 6, session 3, call 7: wave-level counters of the W1|W3-shaped GEMM, row-major operands against the pair layout (and the vendor kernel)
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
export PMC_PASSES="p1 p4 p6"
PMC_OUT=$R/gpurun_out/r6/call11/rowmajor LT_PMC_PAIR=0 bash scripts/gpu_pmc_gemm_stalls.sh
PMC_OUT=$R/gpurun_out/r6/call11/pair LT_PMC_PAIR=1 LT_PMC_VENDOR=0 bash scripts/gpu_pmc_gemm_stalls.sh
echo "== row-major + vendor"; cat $R/gpurun_out/r6/call11/rowmajor/summary.txt | cut -c1-330
echo "== pair"; cat $R/gpurun_out/r6/call11/pair/summary.txt | cut -c1-330
