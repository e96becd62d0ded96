#!/bin/bash
# GPU-box helper: route 1b (sample + collect) against route 1a (full select) for large k.
cd "$(dirname "$0")/.."
for v in "$@"; do
  echo "=== variant: $v"
  rm -f cellcomm_amd/csrc/knn.o
  if ! make -C cellcomm_amd/csrc -j8 EXTRA="$v" >gpurun_out/sweep_build.log 2>&1; then tail -5 gpurun_out/sweep_build.log; continue; fi
  for k in 64 100 128; do
    for n in 700 5000 30000; do
      python tools/knn_bench.py $n 64 $k $n 3 2>&1 | grep -v amdgpu
    done
  done
done
python -m pytest tests/test_gpu_knn.py tests/test_gpu_fuzz.py -m gpu -x -q 2>&1 | tail -3
