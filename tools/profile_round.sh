#!/bin/bash
# GPU-box helper: the evidence committed under profiles/ for a round (round 2 on).  Everything lands in
# gpurun_out/profiles_<tag>/ with the names it keeps under profiles/:
#   <tag>_bench_kernel_stats.csv / <tag>_bench_line_under_rocprof.json   rocprofv3 --kernel-trace --stats of the default bench command
#                                 + --no-dense-host (the dense-host side numbers launch the same layer-1 kernel on 4096-row
#                                 and 32-row inputs outside the timed region, which would blur the per-kernel average)
#   <tag>_pmc_hbm_traffic.json    FETCH_SIZE / WRITE_SIZE / TCC hit+miss per kernel, separate passes (bench.py replays the dominant kernel's traffic from it)
#   <tag>_pmc_sq_dense.json       SQ counters + clock of the fp32 dense path (one 8192-row pass per launch)
#   <tag>_pmc_sq_sparse.json      ... of the sparse first layer (16384 rows)
#   <tag>_pmc_gemm16_f16.json / _f16x2.json   ... of the fp16 kernels (one 32768-row pass)
#   <tag>_pmc_knn_c3.json / _c4rank.json      ... of the search at 100k x 100k and at one rank's 125k x 1M (+ _c4rank_kstep.json: the collect's other norm form)
set -u
cd "$(dirname "$0")/.."
tag=${1:-r03}
export TMPDIR=/tmp
root=$PWD
dst=$root/gpurun_out/profiles_$tag
rm -rf "$dst"; mkdir -p "$dst"
(cd /tmp && timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$dst/stats" -- python3 $root/bench.py --steps 5 --warmup 1 --no-dense-host >"$dst/${tag}_bench_line_under_rocprof.json" 2>"$dst/bench_stderr.log") || { echo "stats run failed"; tail -5 "$dst/bench_stderr.log"; exit 1; }
cp $(find "$dst/stats" -name '*kernel_stats.csv' | head -1) "$dst/${tag}_bench_kernel_stats.csv"
rm -rf "$dst/stats"
echo "stats done"; head -6 "$dst/${tag}_bench_kernel_stats.csv" | cut -c1-160
K=gemm_bias_act,csr_gather,spmm_bias_act,gemm16_kernel,knn_
# (8192 cells = ONE pass of the default size: the launch shape of the timed 100k-cell run; since round 4 a call of up to two
# default passes runs as one pass of all its rows, so 16384 cells would profile a launch twice as tall)
python3 tools/pmc_run.py "$dst/${tag}_pmc_hbm_traffic.json" $K --groups tcc -- python3 $root/bench.py --cells 8192 --steps 2 --warmup 0 --no-cpu-baseline --no-dense-host --no-parity-check 2>&1 | tail -n 12
(timeout -k 10 300 python3 $root/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-alt --dense-host 2>/dev/null | python3 -c "import json,sys; print(json.dumps(json.loads(sys.stdin.read().strip().splitlines()[-1])['dense_host'], indent=1))" > "$dst/${tag}_dense_host.json") || echo "dense-host run failed"
python3 tools/pmc_run.py "$dst/${tag}_pmc_sq_dense.json" gemm_bias_act,csr_gather --groups sq -- python3 $root/bench.py --cells 8192 --steps 2 --warmup 0 --no-cpu-baseline --no-alt --no-parity-check 2>&1 | tail -n 5
python3 tools/pmc_run.py "$dst/${tag}_pmc_sq_sparse.json" spmm_bias_act,gemm_bias_act --groups sq -- python3 $root/bench.py --sparse --cells 16384 --steps 2 --warmup 0 --no-cpu-baseline --no-alt --no-parity-check 2>&1 | tail -n 5
for dt in f16 f16x2; do
  python3 tools/pmc_run.py "$dst/${tag}_pmc_gemm16_$dt.json" gemm16_kernel --groups sq -- python3 $root/bench.py --dtype $dt --cells 32768 --steps 2 --warmup 0 --no-cpu-baseline --no-alt --no-parity-check 2>&1 | tail -n 3
done
python3 tools/pmc_run.py "$dst/${tag}_pmc_knn_c3.json" knn_ --groups sq -- python3 $root/tools/knn_bench.py 100000 64 15 100000 2 2>&1 | tail -n 8
python3 tools/pmc_run.py "$dst/${tag}_pmc_knn_c4rank.json" knn_ --groups sq -- python3 $root/tools/knn_bench.py 1000000 64 15 125000 1 2>&1 | tail -n 8
# the same search with the collect's norm K-step form forced (the library picks the accumulator form at this shape since round 4)
CC_KNN_COLLECT_NORM_FORM=1 python3 tools/pmc_run.py "$dst/${tag}_pmc_knn_c4rank_kstep.json" knn_collect --groups sq -- python3 $root/tools/knn_bench.py 1000000 64 15 125000 1 2>&1 | tail -n 3
ls -la "$dst"
