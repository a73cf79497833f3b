#!/bin/bash
# 4-bit formats: skinny streamer vs 64-row tiles at small decode batches (planner threshold)
set -u
for wl in mixtral8x7b_mxfp4_decode_m32 mixtral8x7b_int4g128_decode_m128 mixtral8x7b_nvfp4_decode_m128; do
for m in 8 16 32 64; do
  echo "== $wl M=$m"
  timeout 300 python tools/sweep.py --workload $wl --M $m --reps 20 --cfgs ";;tiled=64;tiled=64,pd1=4" 2>&1 | grep "^\[" | tail -n +2 | cut -c1-135
done
done
