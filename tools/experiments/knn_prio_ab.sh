# s_setprio in the collect's straight path: A = 1 over the MFMA cluster, B = 2 over the tree, C = both (built from patched copies of knn.hip)
cd /root/repo
tools/experiments/ab_lib.sh "python tools/knn_bench.py 1000000 64 15 125000 5; python tools/knn_bench.py 500000 64 50 62500 10; python tools/knn_bench.py 100000 64 15 100000 20" product tools/experiments/_prioA.so tools/experiments/_prioB.so tools/experiments/_prioC.so product
