# the collect's re-check of parked pairs: fine_dist (product since the round's last session) against fine_dist_batched (_batched.so =
# the library before that change), alternating on one box
cd /root/repo
tools/experiments/ab_lib.sh "python tools/knn_bench.py 1000000 64 15 125000 5; python tools/knn_bench.py 500000 64 50 62500 10; python tools/knn_bench.py 100000 64 15 100000 20; python tools/knn_bench.py 100000 32 15 100000 20" product tools/experiments/_batched.so product tools/experiments/_batched.so
