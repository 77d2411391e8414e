#!/bin/bash
# round-2 closing measurements: rocprofv3 kernel stats + PMC of the headline bench, the other BASELINE configs, cfg 1 / cfg 3 kernel stats
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
bash scripts/gpu_prof.sh 2>&1 | tail -40
python scripts/bench_configs.py cfg3 cfg4 2>&1 | grep -v amdgpu.ids | tail -3
python scripts/bench_configs.py cfg1 cfg5 2>&1 | grep -v amdgpu.ids | tail -3
python scripts/bench_configs.py cfg1 cfg5 --pairs 4 2>&1 | grep -v amdgpu.ids | tail -3
bash scripts/gpu_prof_cfg.sh cfg1 2>&1 | tail -16
bash scripts/gpu_prof_cfg.sh cfg3 2>&1 | tail -12
