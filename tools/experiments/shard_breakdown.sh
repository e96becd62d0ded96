for c in 12500 12288 8192 16384; do python bench.py --cells $c --steps 10 --warmup 3 --no-cpu-baseline --no-alt --no-parity-check 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$c cells: %.2f ms/step, host encode %.2f' % (d['ms_per_step'], d['host_stage_ms_rank0']['encode_ms']), {k:(v['launches'],round(v['ms_per_step'],2)) for k,v in d['stages_rank0'].items() if k.startswith('g')})"; done
