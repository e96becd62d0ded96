python tools/shard_step.py f32 2>/dev/null | grep "P=8\|P=1:"
python bench.py --cells 10000 --genes 20000 --latent 32 --steps 30 --warmup 5 --no-cpu-baseline --no-alt --no-parity-check 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('configs[1]: %.0f cells/s, %.2f ms/step, encode %.2f ms' % (d['value'], d['ms_per_step'], d['host_stage_ms_rank0']['encode_ms']))"
for c in 12500 10000 14000 16384; do python bench.py --cells $c --steps 10 --warmup 3 --no-cpu-baseline --no-alt --no-parity-check 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$c cells @30000: %.0f cells/s, %.2f ms/step, encode %.2f ms' % (d['value'], d['ms_per_step'], d['host_stage_ms_rank0']['encode_ms']))"; done
