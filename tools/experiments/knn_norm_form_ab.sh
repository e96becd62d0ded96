set -e
cd /root/repo
timeout -k 10 500 python -m pytest tests/test_gpu_knn.py -x -q -k "norm_forms or coarse_and_fp32 or collect_groups" > gpurun_out/nf_tests.log 2>&1 || { tail -30 gpurun_out/nf_tests.log; exit 1; }
tail -2 gpurun_out/nf_tests.log
for rep in 1 2; do
for form in 1 2; do
  echo "== form $form"
  CC_KNN_COLLECT_NORM_FORM=$form timeout -k 10 120 python tools/knn_bench.py 1000000 64 15 125000 5 2>&1 | grep -v amdgpu.ids
  CC_KNN_COLLECT_NORM_FORM=$form timeout -k 10 120 python tools/knn_bench.py 500000 64 50 62500 10 2>&1 | grep -v amdgpu.ids
  CC_KNN_COLLECT_NORM_FORM=$form timeout -k 10 120 python tools/knn_bench.py 100000 64 15 100000 20 2>&1 | grep -v amdgpu.ids
done
done
