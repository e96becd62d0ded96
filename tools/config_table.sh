#!/bin/bash
# GPU-box helper: one bench line per BASELINE.json config shape that fits one GPU (for the table in DESIGN.md).
cd "$(dirname "$0")/.."
run() {
  echo "--- $*"
  timeout -k 10 400 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-alt "$@" | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); s = d['stages_rank0']
print(round(d['value']), 'cells/s', round(d['ms_per_step'],2), 'ms', d['dtype'], {k: round(v['ms_per_step'],2) for k,v in s.items()})"
}
run --cells 10000 --genes 20000 --latent 32 --k 15                      # configs[1] (encode-only in BASELINE; kNN added)
run --cells 100000 --genes 30000 --latent 64 --k 15                     # configs[2]
run --cells 100000 --genes 30000 --latent 64 --k 15 --arch cls          # same size, classify encoder
run --cells 125000 --genes 30000 --latent 64 --k 15                     # one of 8 shards of configs[3] (kNN against own shard only)
run --cells 62500 --genes 33000 --latent 64 --k 50 --dtype f16          # one of 8 shards of configs[4]
