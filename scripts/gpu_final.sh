#!/bin/bash
# Round-end validation in one gpurun call: every -m gpu test (incl. the opt-in full-2B parity test), smoke(), the default
# bench line, then the rocprofv3 kernel-trace + PMC passes (scripts/gpu_prof.sh).  Logs -> gpurun_out/.
set -u
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
echo "== nproc $(nproc)"
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { echo BUILD FAILED; tail -30 gpurun_out/build.log; }
tail -1 gpurun_out/build.log
echo "== pytest -m gpu (all files)"
LUMINA_SLOW_TESTS=${SLOW:-1} timeout 1500 python -m pytest tests -m gpu -q --timeout=900 -p no:cacheprovider -rs > gpurun_out/pytest_gpu.log 2>&1
echo "exit $?"; tail -8 gpurun_out/pytest_gpu.log
echo "== smoke"
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "exit $?"; tail -3 gpurun_out/smoke.log
echo "== bench (default flags)"
timeout 900 python bench.py > gpurun_out/bench.log 2>&1; echo "exit $?"; tail -1 gpurun_out/bench.log | cut -c1-1500
echo "== rocprofv3"
PROF_STEPS=6 bash scripts/gpu_prof.sh
