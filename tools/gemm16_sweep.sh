#!/bin/bash
# GPU-box helper: rebuild gemm16 with different knobs, time f16x2 and f16 encoder stages.
set -u
cd "$(dirname "$0")/.."
for v in "$@"; do
  echo "=== variant: $v"
  rm -f cellcomm_amd/csrc/gemm16.o
  if ! make -C cellcomm_amd/csrc -j8 EXTRA="$v" >gpurun_out/sweep_build.log 2>&1; then tail -5 gpurun_out/sweep_build.log; continue; fi
  for dt in f16x2 f16; do
  timeout -k 10 200 python bench.py --cells ${SWEEP_CELLS:-32768} --steps 3 --warmup 1 --no-cpu-baseline --no-alt --dtype $dt 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); s=d['stages_rank0']
        print('$dt cells/s %.0f | L1 %.1f TF | L2 %.1f TF' % (d['value'], s['gemm_l1']['rate']/1e12, s['gemm_l2']['rate']/1e12))
"
  done
done
