for br in 0 4200 3200 2100 1600; do
  python bench.py --cells 12500 --steps 10 --warmup 2 --no-cpu-baseline --no-alt --batch-rows $br | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); s = d['stages_rank0']
print('br=$br', round(d['ms_per_step'],2), {k: round(v['ms_per_step'],2) for k,v in s.items()})"
done
