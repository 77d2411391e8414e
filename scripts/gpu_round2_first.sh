#!/bin/bash
# First GPU call of round 2 (~10 GPU-minutes): the three measurements DESIGN.md 5.1 asks for before any more GEMM code is written.
#   1. scripts/ubench/gemm_energy.hip  - what each main-loop ingredient costs under the power-managed clock, 1 vs 2 waves per SIMD
#   2. scripts/gemm_trace.py           - s_memtime breakdown of the 4-wave VGPR-staged kernel (variant 12) next to the 8-wave ping-pong
#   3. scripts/opbench.py attn_vendor - library attention (torch SDPA) on the engine's shape, next to attention_variant 3
#   4. scripts/gpu_prof_vendor.sh      - PMC passes (clock = GRBM_GUI_ACTIVE / time, duty = MFMA busy / cycles) incl. variants 10 and 12
#   usage: gpurun --timeout 900 -- 'bash scripts/gpu_round2_first.sh'
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/round2_first
mkdir -p $OUT
cd $R
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-value scripts/ubench/gemm_energy.hip -o /tmp/gemm_energy > $OUT/build.log 2>&1
timeout 120 /tmp/gemm_energy > $OUT/gemm_energy.log 2>&1; echo "gemm_energy exit $?"; cat $OUT/gemm_energy.log
LT_PP_TAIL=0 timeout 300 python scripts/gemm_trace.py > $OUT/gemm_trace.log 2>&1; echo "gemm_trace exit $?"; grep -v "^  blk  *\(6[0-9]\|1[0-9][0-9]\)" $OUT/gemm_trace.log | head -60
LUMINA_EXPERIMENTAL=1 timeout 300 python -m pytest tests/test_gpu_ops.py -q -k "gemm or specialised" > $OUT/pytest_gemm.log 2>&1; echo "pytest gemm exit $?"; tail -2 $OUT/pytest_gemm.log
LUMINA_EXPERIMENTAL=1 timeout 300 python -m pytest tests/test_gpu_model.py tests/test_gpu_variants.py -q -k "cli or sample_driver" > $OUT/pytest_cli.log 2>&1; echo "pytest drivers exit $?"; tail -2 $OUT/pytest_cli.log
timeout 300 python scripts/opbench.py gemm --rounds 3 --gemm-variants 3,10,13,14 > $OUT/opbench_gemm.log 2>&1; echo "opbench exit $?"; tail -12 $OUT/opbench_gemm.log
# the same with the CUs of an XCD spread over eight tile phases (is the chip-wide synchronised prologue / epilogue what holds the clock down?)
timeout 300 python scripts/opbench.py gemm --rounds 3 --gemm-variants 10,13,14 --gemm-stagger 2 > $OUT/opbench_gemm_stagger2.log 2>&1; echo "opbench stagger exit $?"; tail -9 $OUT/opbench_gemm_stagger2.log
timeout 300 python scripts/opbench.py elem --rounds 5 > $OUT/opbench_elem.log 2>&1; echo "opbench elem exit $?"; tail -9 $OUT/opbench_elem.log
timeout 300 python scripts/opbench.py attn_vendor --rounds 3 > $OUT/opbench_attn_vendor.log 2>&1; echo "attn_vendor exit $?"; tail -9 $OUT/opbench_attn_vendor.log
bash scripts/gpu_prof_vendor.sh > $OUT/prof_vendor.log 2>&1; echo "prof_vendor exit $?"; tail -12 $OUT/prof_vendor.log
