python tools/shard_step.py f32 2>/dev/null | grep "P="
python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-alt --no-parity-check 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('C3: %.0f cells/s, %.2f ms/step' % (d['value'], d['ms_per_step']))"
