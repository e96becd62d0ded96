# the tree's library against tools/experiments/_prev.so (the library of the previous commit), alternating on one box
cd /root/repo
tools/experiments/ab_lib.sh "python tools/knn_bench.py 1000000 64 15 125000 5; python tools/knn_bench.py 500000 64 50 62500 10; python tools/knn_bench.py 100000 64 15 100000 20; python tools/knn_bench.py 100000 32 15 100000 20" product tools/experiments/_prev.so product tools/experiments/_prev.so
