cd /root/repo
for br in 8192 12288 24576; do
  timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-alt --arch cls --batch-rows $br | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); s = d['stages_rank0']
print('cls br=$br', round(d['value']), round(d['ms_per_step'],2), {k: round(v['ms_per_step'],2) for k,v in s.items()})"
done
