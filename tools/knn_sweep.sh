#!/bin/bash
# GPU-box helper: rebuild with different kNN collect knobs and time the kNN stages.
set -u
cd "$(dirname "$0")/.."
for v in "$@"; do
  echo "=== variant: $v"
  rm -f cellcomm_amd/csrc/knn.o
  if ! make -C cellcomm_amd/csrc -j8 EXTRA="$v" >gpurun_out/sweep_build.log 2>&1; then tail -5 gpurun_out/sweep_build.log; continue; fi
  python tools/knn_bench.py 100000 64 15 2>&1 | grep -v amdgpu
  python tools/knn_bench.py 100000 64 15 12500 2>&1 | grep -v amdgpu
done
