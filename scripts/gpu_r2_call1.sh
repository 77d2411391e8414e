#!/bin/bash
# round 2, GPU call 1: first hardware run of the experimental persistent 4-wave GEMMs (variants 13 / 14) next to the defaults,
# the mode-specialised gated_residual_norm, and the row-kernel baseline (trimmed form of scripts/gpu_round2_first.sh)
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r2/call1
mkdir -p $OUT
cd $R
LUMINA_EXPERIMENTAL=1 timeout 240 python -m pytest tests/test_gpu_ops.py -q -x -k "gemm or specialised" > $OUT/pytest_gemm.log 2>&1; echo "pytest gemm exit $?"; tail -3 $OUT/pytest_gemm.log
timeout 200 python scripts/opbench.py gemm --rounds 3 --gemm-variants 0,3,10,13,14 > $OUT/opbench_gemm.log 2>&1; echo "opbench exit $?"; tail -26 $OUT/opbench_gemm.log
timeout 120 python scripts/opbench.py gemm --rounds 3 --gemm-variants 10,13,14 --gemm-stagger 2 > $OUT/opbench_gemm_stagger2.log 2>&1; echo "opbench stagger exit $?"; tail -4 $OUT/opbench_gemm_stagger2.log
timeout 120 python scripts/opbench.py elem --rounds 5 > $OUT/opbench_elem.log 2>&1; echo "opbench elem exit $?"; tail -12 $OUT/opbench_elem.log
python - <<'PY' > $OUT/hostinfo.log 2>&1
import os, torch, importlib.util
print("cpus", os.cpu_count(), "gpus", torch.cuda.device_count(), torch.cuda.get_device_name(0))
print("torchdiffeq", importlib.util.find_spec("torchdiffeq"))
print(open("/proc/meminfo").read().split("\n")[0])
PY
cat $OUT/hostinfo.log
