#!/bin/bash
# per-kernel times of one knn_bench shape under rocprofv3 (GPU box): tools/experiments/knn_kstats.sh <tag> N d k nq reps
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kst_$tag
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kst_$tag -- python3 /root/repo/tools/knn_bench.py "$@" > /tmp/kst_$tag.log 2>&1
python3 - "$tag" <<'PY'
import csv, glob, sys
tag = sys.argv[1]
f = glob.glob(f'/tmp/kst_{tag}/**/*kernel_stats.csv', recursive=True)
rows = list(csv.DictReader(open(f[0])))
for r in rows[:14]:
    print(tag, r['Name'][:60], 'calls', r['Calls'], 'avg_us %.1f' % (float(r['AverageNs']) / 1e3))
PY
