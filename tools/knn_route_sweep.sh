#!/bin/bash
# GPU-box helper: where route 1b (sample + collect) starts to beat route 1a (full select).
cd "$(dirname "$0")/.."
for v in "$@"; do
  echo "=== variant: $v"
  rm -f cellcomm_amd/csrc/knn.o
  if ! make -C cellcomm_amd/csrc -j8 EXTRA="$v" >gpurun_out/sweep_build.log 2>&1; then tail -5 gpurun_out/sweep_build.log; continue; fi
  for n in 200 300 600 1000 1500 2048; do
    python tools/knn_bench.py $n 64 15 $n 5 2>&1 | grep -v amdgpu
    python tools/knn_bench.py $n 64 50 $n 5 2>&1 | grep -v amdgpu
  done
done
