#!/bin/bash
# A/B of two builds of the library (LKM_LIB_PATH): bash tools/gpu_ab.sh "<libA> <libB>" "<workload:M:cfgs> ..."
set -u
LIBS=${1:-"liblkm.so"}
shift
for rep in 1 2; do
for lib in $LIBS; do
  for spec in "$@"; do
    IFS=: read -r wl m cfgs <<< "$spec"
    echo "== $lib $wl M=$m"
    LKM_LIB_PATH=$PWD/lvllm_amd/$lib timeout 300 python tools/sweep.py --workload $wl --M $m --reps 20 --cfgs ";$cfgs" 2>&1 | grep "^\[" | tail -n +2 | cut -c1-150
  done
done
done
