#!/bin/bash
set -u
timeout 600 python -m pytest tests/test_gpu_moe.py -m gpu -x -q -k "prefetch or int4 or fp4 or quantised" 2>&1 | tail -2
for wl in mixtral8x7b_int4g128_decode_m128 mixtral8x7b_mxfp4_decode_m128 mixtral8x7b_nvfp4_decode_m128; do
  echo "== $wl"
  timeout 300 python tools/sweep.py --workload $wl --reps 20 --cfgs ";;pd1=4;pd1=8;pd2=2;pd2=8;pd1=8,pd2=8;pd1=2,pd2=4" 2>&1 | grep "^\[" | tail -n +2 | cut -c1-135
done
echo "== mxfp4 M=32"; timeout 300 python tools/sweep.py --workload mixtral8x7b_mxfp4_decode_m32 --reps 20 --cfgs ";;tiled=64;tiled=64,pd1=8,pd2=8;tiled=64,pd1=4" 2>&1 | grep "^\[" | tail -n +2 | cut -c1-135
