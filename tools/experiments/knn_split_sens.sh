# per-workgroup overhead of the collect: the same search with the candidate range cut into S pieces
for s in 8 16 24 32 48 64; do
  echo -n "C3 S=$s: "; CC_KNN_COLLECT_SPLITS=$s python tools/knn_bench.py 100000 64 15 100000 10 2>/dev/null | grep -o "knn_select': [0-9.]*"
done
for s in 8 16 32 64; do
  echo -n "c5rank S=$s: "; CC_KNN_COLLECT_SPLITS=$s python tools/knn_bench.py 500000 64 50 62500 5 2>/dev/null | grep -o "knn_select': [0-9.]*"
done
for s in 16 32 64; do
  echo -n "c4rank S=$s: "; CC_KNN_COLLECT_SPLITS=$s python tools/knn_bench.py 1000000 64 15 125000 3 2>/dev/null | grep -o "knn_select': [0-9.]*"
done
python tools/experiments/knn_host_overhead.py 2>&1 | grep -v amdgpu
