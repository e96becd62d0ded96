#!/bin/bash
# GPU-box helper: the part of tools/profile_round.sh that a change to the search's small kernels makes stale — the
# bench line + kernel stats under rocprofv3 and the two kNN counter files.  Same names, same destination.
set -u
cd "$(dirname "$0")/.."
tag=${1:-r04}
export TMPDIR=/tmp CC_KNN_HOST_LEG=0
root=$PWD
dst=$root/gpurun_out/profiles_$tag
rm -rf "$dst"; mkdir -p "$dst"
(cd /tmp && timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$dst/stats" -- python3 $root/bench.py --steps 5 --warmup 1 --no-dense-host >"$dst/${tag}_bench_line_under_rocprof.json" 2>"$dst/bench_stderr.log") || { echo "stats run failed"; tail -5 "$dst/bench_stderr.log"; exit 1; }
cp $(find "$dst/stats" -name '*kernel_stats.csv' | head -1) "$dst/${tag}_bench_kernel_stats.csv"
rm -rf "$dst/stats"
echo "stats done"; head -4 "$dst/${tag}_bench_kernel_stats.csv" | cut -c1-160
python3 tools/pmc_run.py "$dst/${tag}_pmc_knn_c3.json" knn_ --groups sq -- python3 $root/tools/knn_bench.py 100000 64 15 100000 2 2>&1 | tail -n 8
python3 tools/pmc_run.py "$dst/${tag}_pmc_knn_c4rank.json" knn_ --groups sq -- python3 $root/tools/knn_bench.py 1000000 64 15 125000 1 2>&1 | tail -n 8
ls -la "$dst"
