#!/bin/bash
# GPU-box helper: rebuild gemm16 with each flag set, time the fp16 encoder layers alone (tools/experiments/g16_ablate.py).
#   tools/experiments/g16_ablate.sh "<flags 1>" "<flags 2>" ...     ABL_SHAPE="32768 30000 f16"
# The switches live in g16_ablations.patch (round 4: the product source has none): the patch is applied to the scratch
# copy on the GPU box for the duration of the sweep and taken out again.  It applies to csrc/gemm16.hip as of commit 8cf6e37
# (before the A16T operand layout): `git show 8cf6e37:cellcomm_amd/csrc/gemm16.hip > cellcomm_amd/csrc/gemm16.hip` first.  NEVER pass -DCC_PP_NOMFMA (see the patch).
set -u
cd "$(dirname "$0")/../.."
for v in "$@"; do case "$v" in *CC_PP_NOMFMA*) echo "refusing -DCC_PP_NOMFMA: it faults the GPU (tools/experiments/README.md)"; exit 2;; esac; done
patch -p1 < tools/experiments/g16_ablations.patch || exit 1
trap 'patch -R -p1 < tools/experiments/g16_ablations.patch; rm -f cellcomm_amd/csrc/gemm16.o; make -C cellcomm_amd/csrc -j8 >gpurun_out/abl_build.log 2>&1' EXIT
shape=${ABL_SHAPE:-32768 30000 f16}
for v in "$@"; do
  rm -f cellcomm_amd/csrc/gemm16.o
  if ! make -C cellcomm_amd/csrc -j8 EXTRA="$v" >gpurun_out/abl_build.log 2>&1; then tail -5 gpurun_out/abl_build.log; continue; fi
  printf "%-40s " "'$v'"
  timeout -k 10 120 python tools/experiments/g16_ablate.py $shape 2>/dev/null
done
