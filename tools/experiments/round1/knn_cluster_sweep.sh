cd /root/repo
for v in "" "-DCC_KNN_SAMPLE=16 -DCC_KNN_SAMPLE_SPARE=32"; do
  echo "=== $v"; rm -f cellcomm_amd/csrc/knn.o; make -C cellcomm_amd/csrc -j8 EXTRA="$v" >gpurun_out/sweep_build.log 2>&1
  python tools/knn_cluster_bench.py 2>&1 | grep -v amdgpu
done
