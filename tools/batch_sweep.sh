#!/bin/bash
# GPU-box helper: rows per encoder pass vs step time for one dtype.
cd "$(dirname "$0")/.."
dt=${1:-f16x2}; shift
for br in "$@"; do
  timeout -k 10 300 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-alt --dtype $dt --batch-rows $br | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); s = d['stages_rank0']
print('$dt br=$br', round(d['value']), round(d['ms_per_step'],2), {k: round(v['ms_per_step'],2) for k,v in s.items()})"
done
