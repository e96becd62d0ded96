#!/bin/bash
# GPU-box helper: the evidence committed under profiles/ for a round.
#   1. rocprofv3 --kernel-trace --stats of the default bench command (same command whose JSON line is the headline)
#   2. FETCH_SIZE / WRITE_SIZE / TCC_HIT / TCC_MISS per kernel, separate passes, one 8192-row pass per launch
#   3. SQ busy counters of the same launch shape
# Everything lands in gpurun_out/profiles_<tag>/.
set -u
cd "$(dirname "$0")/.."
tag=${1:-r01}
export TMPDIR=/tmp
root=$PWD
dst=$root/gpurun_out/profiles_$tag
rm -rf "$dst"; mkdir -p "$dst"
(cd /tmp && timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$dst/stats" -- python3 $root/bench.py --steps 5 --warmup 1 >"$dst/bench_line_under_rocprof.json" 2>"$dst/bench_stderr.log") || { echo "stats run failed"; tail -5 "$dst/bench_stderr.log"; exit 1; }
cp $(find "$dst/stats" -name '*kernel_stats.csv' | head -1) "$dst/bench_kernel_stats.csv"
echo "stats done"; head -4 "$dst/bench_kernel_stats.csv" | cut -c1-200
tools/pmc_collect.sh $tag FETCH_SIZE WRITE_SIZE TCC_HIT_sum TCC_MISS_sum | grep -E "gemm_bias|gather|failed|done"
cp gpurun_out/pmc_$tag.json "$dst/pmc_hbm_traffic.json"
out=$dst/sq
mkdir -p "$out"
(cd /tmp && timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d "$out" -- python3 $root/bench.py --cells 8192 --steps 2 --warmup 0 --no-cpu-baseline --no-alt >"$out/run.log" 2>&1) || { echo "sq pass failed"; tail -5 "$out/run.log"; }
python3 - "$tag" <<'PY'
import csv, glob, collections, json, sys
tag = sys.argv[1]
base = f'gpurun_out/profiles_{tag}'
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(base + '/sq/**/*counter_collection.csv', recursive=True):
    for row in csv.DictReader(open(f)):
        acc[row['Kernel_Name']][row['Counter_Name']].append(float(row['Counter_Value']))
dur = collections.defaultdict(list)
for f in glob.glob(base + '/sq/**/*kernel_trace.csv', recursive=True):
    for row in csv.DictReader(open(f)):
        dur[row['Kernel_Name']].append((int(row['End_Timestamp']) - int(row['Start_Timestamp'])) / 1e6)
res = {}
for k, v in acc.items():
    if not any(t in k for t in ('gemm', 'gather', 'knn')): continue
    r = {c: {'launches': len(x), 'mean_per_launch': sum(x) / len(x)} for c, x in v.items()}
    ms = sum(dur[k]) / max(len(dur[k]), 1)
    cyc = r.get('GRBM_GUI_ACTIVE', {}).get('mean_per_launch', 0) / 8
    r['ms'] = ms
    if cyc and ms:
        r['clock_GHz'] = cyc / ms / 1e6
        r['mfma_pipe_busy_frac'] = r.get('SQ_VALU_MFMA_BUSY_CYCLES', {}).get('mean_per_launch', 0) / (1024 * cyc)
    res[k] = r
json.dump({'command': 'rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT --kernel-trace -- python3 bench.py --cells 8192 --steps 2 --warmup 0 --no-cpu-baseline --no-alt',
           'note': 'means per launch; clock = GRBM_GUI_ACTIVE/8/duration; mfma_pipe_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x cycles)',
           'kernels': res}, open(base + '/pmc_sq.json', 'w'), indent=1)
for k, v in res.items():
    if v['ms'] > 0.3: print(k[:50], 'ms=%.2f clk=%.2f mfma_busy=%.3f' % (v['ms'], v.get('clock_GHz', 0), v.get('mfma_pipe_busy_frac', 0)))
PY
rm -rf "$dst/stats" "$dst/sq"
ls -la "$dst"
