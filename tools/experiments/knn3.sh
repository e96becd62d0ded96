python tools/knn_bench.py 100000 64 15 100000 10; python tools/knn_bench.py 1000000 64 15 125000 3; python tools/knn_bench.py 500000 64 50 62500 5
