#!/bin/bash
# GPU-box helper (round 3): where the coarse collect's time goes at one rank's shape of configs[3] and at C3.
# Applies tools/experiments/knn_collect_c_operand_and_ablations.patch to a scratch copy of knn.hip is NOT done here: the
# ablation hooks (CC_KNN_ABL_NOMASK / _NOBAR / _NOREAD) live in that patch, not in the product source; apply it first
# (git apply tools/experiments/knn_collect_c_operand_and_ablations.patch), run this, then git checkout the file.
# Builds knn.hip with an ablation macro, times tools/knn_bench.py, restores the product build.  Results of the ablated
# builds are WRONG by construction (pairs are dropped); only the kernel times are read.
set -u
cd "$(dirname "$0")/../.."
variants=("" "-DCC_KNN_ABL_NOMASK" "-DCC_KNN_ABL_NOMASK -DCC_KNN_ABL_NOBAR" "-DCC_KNN_ABL_NOMASK -DCC_KNN_ABL_NOREAD" "-DCC_KNN_ABL_NOMASK -DCC_KNN_ABL_NOREAD -DCC_KNN_ABL_NOBAR")
for v in "${variants[@]}"; do
  echo "=== variant: '$v'"
  rm -f cellcomm_amd/csrc/knn.o
  make -C cellcomm_amd/csrc -j8 EXTRA="$v" >gpurun_out/abl_build.log 2>&1 || { tail -5 gpurun_out/abl_build.log; continue; }
  python tools/knn_bench.py 1000000 64 15 125000 1 2>/dev/null | tail -1
  python tools/knn_bench.py 100000 64 15 100000 2 2>/dev/null | tail -1
done
rm -f cellcomm_amd/csrc/knn.o
make -C cellcomm_amd/csrc -j8 >gpurun_out/abl_build.log 2>&1 || tail -5 gpurun_out/abl_build.log
