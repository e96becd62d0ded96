#!/bin/bash
# GPU-box helper: rebuild gemm16 with different knobs, time the fp16 encoder layers.
#   tools/gemm16_sweep.sh "<EXTRA flags of variant 1>" "<variant 2>" ...      ("" = the product build)
# SWEEP_SHAPES: bench flag sets, ';'-separated (default: one 32768-row pass at C3's width, and configs[4]'s rank shape)
# Variants with -DCC_G16_* / -DCC_PP_* flags need tools/experiments/g16_ablations.patch (SWEEP_PATCH=1 applies it for the
# run; the product source has no build-time switches since round 4).
set -u
cd "$(dirname "$0")/.."
if [ "${SWEEP_PATCH:-0}" = 1 ]; then
  patch -p1 < tools/experiments/g16_ablations.patch || exit 1
  trap 'patch -R -p1 < tools/experiments/g16_ablations.patch' EXIT
fi
shapes=${SWEEP_SHAPES:---cells 32768;--cells 62500 --genes 33000 --k 50}
dtypes=${SWEEP_DTYPES:-f16 f16x2}
for v in "$@"; do
  echo "=== variant: '$v'"
  rm -f cellcomm_amd/csrc/gemm16.o
  if ! make -C cellcomm_amd/csrc -j8 EXTRA="$v" >gpurun_out/sweep_build.log 2>&1; then tail -5 gpurun_out/sweep_build.log; continue; fi
  IFS=';' read -ra SH <<< "$shapes"
  for sh in "${SH[@]}"; do
    for dt in $dtypes; do
      timeout -k 10 200 python bench.py $sh --steps 5 --warmup 2 --no-cpu-baseline --no-alt --dtype $dt 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); s=d['stages_rank0']
        print('$sh $dt: %.0f cells/s | L1 %.1f TF (%.3f ms/launch) | L2 %.1f TF' % (d['value'], s['gemm_l1']['rate']/1e12, s['gemm_l1']['ms_per_step']/s['gemm_l1']['launches']*d['steps'], s['gemm_l2']['rate']/1e12))
"
    done
  done
done
rm -f cellcomm_amd/csrc/gemm16.o
make -C cellcomm_amd/csrc -j8 >gpurun_out/sweep_build.log 2>&1 || tail -5 gpurun_out/sweep_build.log
