#!/bin/bash
# GPU-box helper: effect of 16-register-aligned MFMA accumulators (asm pin) on the kNN kernels and the fp16 GEMM.
cd "$(dirname "$0")/.."
for v in "$@"; do
  echo "=== variant: $v"
  rm -f cellcomm_amd/csrc/knn.o cellcomm_amd/csrc/gemm16.o
  if ! make -C cellcomm_amd/csrc -j8 EXTRA="$v" >gpurun_out/sweep_build.log 2>&1; then tail -5 gpurun_out/sweep_build.log; continue; fi
  python tools/knn_bench.py 100000 64 15 2>&1 | grep -v amdgpu
  CC_FULL=1 python tools/knn_bench.py 100000 64 15 100000 1 2>&1 | grep -v amdgpu
  timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-alt --dtype f16x2 | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); s = d['stages_rank0']
print('f16x2', round(d['value']), round(d['ms_per_step'],2), {k: round(v['ms_per_step'],2) for k,v in s.items()})"
done
