#!/bin/bash
# GPU-box helper: per-kernel PMC counters of a short bench run, one rocprofv3 pass per counter
# (counters only, no trace domains besides --kernel-trace), summarised to gpurun_out/pmc_<tag>.json.
#   tools/pmc_collect.sh <tag> [counter ...]        default counters: FETCH_SIZE WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
set -u
cd "$(dirname "$0")/.."
tag=$1; shift
counters=("$@")
[ ${#counters[@]} -eq 0 ] && counters=(FETCH_SIZE WRITE_SIZE TCC_HIT_sum TCC_MISS_sum)
export TMPDIR=/tmp
root=$PWD
for c in "${counters[@]}"; do
  out=$root/gpurun_out/pmc_${tag}/$c
  rm -rf "$out"; mkdir -p "$out"
  (cd /tmp && timeout -k 10 400 rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$out" -- python3 $root/bench.py --cells 8192 --steps 2 --warmup 0 --no-cpu-baseline --no-alt >"$out/bench.log" 2>&1) || { echo "pass $c failed"; tail -3 "$out/bench.log"; exit 1; }
  echo "pass $c done"
done
python3 - "$tag" "${counters[@]}" <<'PY'
import csv, glob, json, sys, collections
tag, counters = sys.argv[1], sys.argv[2:]
res = collections.defaultdict(dict)
for c in counters:
    files = glob.glob(f'gpurun_out/pmc_{tag}/{c}/**/*counter_collection.csv', recursive=True)
    acc = collections.defaultdict(list)
    for f in files:
        for row in csv.DictReader(open(f)):
            if row['Counter_Name'] == c:
                acc[row['Kernel_Name']].append(float(row['Counter_Value']))
    for k, v in acc.items():
        res[k][c] = {'launches': len(v), 'mean_per_launch': sum(v) / len(v), 'max': max(v)}
keep = {k: v for k, v in res.items() if 'gemm' in k or 'gather' in k or 'knn' in k}
json.dump({'counters': counters, 'kernels': keep}, open(f'gpurun_out/pmc_{tag}.json', 'w'), indent=1)
for k, v in keep.items():
    line = k[:60].ljust(60)
    for c in counters:
        if c in v: line += f' {c}={v[c]["mean_per_launch"]:.4g}'
    if 'TCC_HIT_sum' in v and 'TCC_MISS_sum' in v:
        h, m = v['TCC_HIT_sum']['mean_per_launch'], v['TCC_MISS_sum']['mean_per_launch']
        line += f' L2hit={h / (h + m):.3f}'
    print(line)
PY
