# one variant's numbers: configs[1] step and a 12.5k-row shard step (run under tools/experiments/ab_lib.sh)
python bench.py --cells 10000 --genes 20000 --latent 32 --steps 30 --warmup 5 --no-cpu-baseline --no-alt --no-parity-check 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('configs[1]: %.0f cells/s, %.2f ms/step, encode %.2f ms' % (d['value'], d['ms_per_step'], d['host_stage_ms_rank0']['encode_ms']))"
SHARD_P=8,4 python tools/shard_step.py f32 2>/dev/null | grep "P="
python bench.py --cells 25000 --steps 5 --warmup 2 --no-cpu-baseline --no-alt --no-parity-check 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('25k cells: %.0f cells/s, %.2f ms/step' % (d['value'], d['ms_per_step']))"
