# needs tools/experiments/knn_collect_qw4.patch applied (the product refuses CC_KNN_OPT_COLLECT_GROUPS = 4)
for g in 0 4 0 4; do
  echo "--- CC_KNN_COLLECT_GROUPS=$g"
  CC_KNN_COLLECT_GROUPS=$g python tools/knn_bench.py 100000 64 15 100000 10
  CC_KNN_COLLECT_GROUPS=$g python tools/knn_bench.py 1000000 64 15 125000 3
  CC_KNN_COLLECT_GROUPS=$g python tools/knn_bench.py 500000 64 50 62500 5
done
