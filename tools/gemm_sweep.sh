#!/bin/bash
# GPU-box helper: rebuild the library with different GEMM knobs and time the encoder stages.
# usage: tools/gemm_sweep.sh "<EXTRA flags variant 1>" "<variant 2>" ...
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
for v in "$@"; do
  echo "=== variant: $v"
  make -C cellcomm_amd/csrc clean >/dev/null
  if ! make -C cellcomm_amd/csrc -j8 EXTRA="$v" >gpurun_out/sweep_build.log 2>&1; then tail -5 gpurun_out/sweep_build.log; continue; fi
  timeout -k 10 200 python bench.py --cells ${SWEEP_CELLS:-24576} --steps 3 --warmup 1 --no-cpu-baseline ${SWEEP_ARGS:-} 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); s=d['stages_rank0']
        print('cells/s %.0f | L1 %.1f TF | L2 %.1f TF | gather %.0f GB/s | knn_select %.2f ms' % (d['value'], s['gemm_l1']['rate']/1e12, s['gemm_l2']['rate']/1e12, s['gather']['rate']/1e9, s['knn_select']['ms_per_step']))
"
done
