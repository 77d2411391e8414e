#!/bin/bash
# One gpurun call worth of checks: op parity, model parity, smoke, short bench.  Logs -> gpurun_out/.
set -u
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
echo "== rocm-smi"; rocm-smi --showproductname 2>/dev/null | head -8
echo "== nproc $(nproc)"
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { echo BUILD FAILED; tail -30 gpurun_out/build.log; }
for f in test_gpu_ops test_gpu_model; do
  echo "== pytest $f"
  timeout 1200 python -m pytest tests/$f.py -m gpu -q --timeout=600 -p no:cacheprovider > gpurun_out/$f.log 2>&1
  echo "exit $?"; tail -40 gpurun_out/$f.log
done
echo "== smoke"
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "exit $?"; tail -5 gpurun_out/smoke.log
echo "== bench"
timeout 900 python bench.py --steps ${BENCH_STEPS:-6} --warmup 2 > gpurun_out/bench.log 2>&1; echo "exit $?"; tail -5 gpurun_out/bench.log
