#!/bin/bash
# GPU-box helper: run one command under several builds of the library, on the same box, back to back.
#   tools/experiments/ab_lib.sh "<command>" <lib1.so> <lib2.so> ...     ("product" = the library in the tree)
# The variant is swapped into cellcomm_amd/libcellcomm_hip.so of the scratch copy for the run and the product restored.
set -u
cd "$(dirname "$0")/../.."
cmd=$1; shift
cp cellcomm_amd/libcellcomm_hip.so /tmp/cc_product.so
trap 'cp /tmp/cc_product.so cellcomm_amd/libcellcomm_hip.so' EXIT
for lib in "$@"; do
  if [ "$lib" = product ]; then cp /tmp/cc_product.so cellcomm_amd/libcellcomm_hip.so; else cp "$lib" cellcomm_amd/libcellcomm_hip.so; fi
  echo "=== $lib"
  timeout -k 10 300 bash -c "$cmd" 2>&1 | grep -v amdgpu.ids
done
