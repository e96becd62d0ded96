cd /root/repo
for nq in 12500 25000 50000 100000; do
  for sp in 8 7 6 0; do
    echo -n "splits=$sp "; CC_KNN_COLLECT_SPLITS=$sp python tools/knn_bench.py 100000 64 15 $nq 5 2>&1 | grep -v amdgpu
  done
done
