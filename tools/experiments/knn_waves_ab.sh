# coarse collect: 4-wave workgroups (3 per CU) against ONE 12-wave workgroup per CU, both norm forms, stage x1 / x2
cd /root/repo
timeout -k 10 500 python -m pytest tests/test_gpu_knn.py -x -q -k "norm_forms" > gpurun_out/wv_tests.log 2>&1 || { tail -30 gpurun_out/wv_tests.log; exit 1; }
tail -2 gpurun_out/wv_tests.log
run() { echo "== waves $1 stage_x $2 form $3 splits $4"; for shape in "1000000 64 15 125000 5" "500000 64 50 62500 10" "100000 64 15 100000 20"; do
  CC_KNN_COLLECT_GROUPS=2 CC_KNN_COLLECT_WAVES=$1 CC_KNN_COLLECT_STAGE_X=$2 CC_KNN_COLLECT_NORM_FORM=$3 CC_KNN_COLLECT_SPLITS=$4 timeout -k 10 120 python tools/knn_bench.py $shape 2>&1 | grep "^n=" | cut -c1-140; done; }
run 4 0 1 0
run 4 0 2 0
run 12 0 1 0
run 12 0 2 0
run 12 2 1 0
run 12 2 2 0
run 12 0 2 8
run 12 0 2 16
run 12 2 2 8
run 12 2 2 16
