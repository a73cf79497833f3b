#!/bin/bash
# A/B of two builds of the library on the int4 decode workloads (LKM_LIB_PATH selects the build)
set -u
W=mixtral8x7b_int4g128_decode_m128
for rep in 1 2; do
for lib in liblkm_old.so liblkm.so; do
  for m in 128 32 512; do
    echo "== $lib M=$m"
    LKM_LIB_PATH=$PWD/lvllm_amd/$lib timeout 300 python tools/sweep.py --workload $W --M $m --reps 20 --cfgs ";;pd1=4" 2>&1 | grep "^\[" | tail -2 | cut -c1-150
  done
done
done
