for st in 0 8 10 12 16; do echo -n "c4rank stride=$st: "; CC_KNN_SAMPLE_STRIDE=$st python tools/knn_bench.py 1000000 64 15 125000 3 2>/dev/null | grep -o "knn_select': [0-9.]*.*"; done
for st in 0 6 8 10 12; do echo -n "c5rank stride=$st: "; CC_KNN_SAMPLE_STRIDE=$st python tools/knn_bench.py 500000 64 50 62500 5 2>/dev/null | grep -o "knn_select': [0-9.]*.*"; done
for st in 0 2 3 4 5; do echo -n "C3 stride=$st: "; CC_KNN_SAMPLE_STRIDE=$st python tools/knn_bench.py 100000 64 15 100000 10 2>/dev/null | grep -o "knn_select': [0-9.]*.*"; done
