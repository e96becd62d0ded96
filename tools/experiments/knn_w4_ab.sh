# QW = 1 collect at four waves per SIMD (launch bounds (256, 4): 128 registers) against the product's three, both norm forms
cd /root/repo
for lib in product tools/experiments/_w4.so; do
for form in 1 2; do
for g in 1 2; do
  echo "== $lib form $form groups $g"
  tools/experiments/ab_lib.sh "CC_KNN_COLLECT_GROUPS=$g CC_KNN_COLLECT_NORM_FORM=$form python tools/knn_bench.py 1000000 64 15 125000 5; CC_KNN_COLLECT_GROUPS=$g CC_KNN_COLLECT_NORM_FORM=$form python tools/knn_bench.py 500000 64 50 62500 10; CC_KNN_COLLECT_GROUPS=$g CC_KNN_COLLECT_NORM_FORM=$form python tools/knn_bench.py 100000 64 15 100000 20" $lib
done; done; done
