#!/bin/bash
# GPU-box helper: layer-1/2 GEMM pacing knobs -> bench stages.
cd "$(dirname "$0")/.."
for v in "$@"; do
  echo "=== variant: $v"
  rm -f cellcomm_amd/csrc/gemm.o cellcomm_amd/csrc/api.o
  if ! make -C cellcomm_amd/csrc -j8 EXTRA="$v" >gpurun_out/sweep_build.log 2>&1; then tail -5 gpurun_out/sweep_build.log; continue; fi
  timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-alt | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); s = d['stages_rank0']
print(round(d['value']), round(d['ms_per_step'],2), 'L1 TF', round(d['roofline']['achieved'],1), {k: round(v['ms_per_step'],2) for k,v in s.items()})"
done
