python bench.py --sparse --cells 65536 --steps 3 --warmup 1 --no-cpu-baseline --no-alt --no-parity-check 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); s=d['stages_rank0']; print('spmm_l1 %.2f ms, spmm_l2 %.2f ms per 65536 cells' % (s['spmm_l1']['ms_per_step'], s['spmm_l2']['ms_per_step']))"
