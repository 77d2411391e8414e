#!/bin/bash
# authoring-container helper: rebuild the in-tree library, then run a command on a GPU box.   scripts/gpu.sh <timeout_s> '<command>' [tag]
set -e
cd "$(dirname "$0")/.."
make -C lumina-t2x_amd/csrc -j8 2>&1 | grep -E "error|Error" && { echo "BUILD FAILED"; exit 1; } || true
tag=${3:-last}
mkdir -p gpurun_out/r3
/usr/local/graft/bin/gpurun --timeout "$1" -- "$2" > gpurun_out/r3/$tag.stdout 2>&1 || true
tail -${TAIL:-40} gpurun_out/r3/$tag.stdout
