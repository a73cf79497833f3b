#!/bin/bash
# tiled GEMM2 split-K at decode batches: 8 experts x 1 token tile x 64 row groups = 512 workgroups only
set -u
for spec in mixtral8x7b_int4g128_decode_m128:128 mixtral8x7b_mxfp4_decode_m128:128 mixtral8x7b_nvfp4_decode_m128:128 mixtral8x7b_mxfp4_decode_m32:32 mixtral8x7b_fp8w8a8_decode_m32:64 mixtral8x7b_bf16_decode_m32:128 dsv3_ep8_rank_fp8w8a8_rows256:256; do
  IFS=: read -r wl m <<< "$spec"
  echo "== $wl M=$m"
  timeout 300 python tools/sweep.py --workload $wl --M $m --reps 16 --cfgs ";;sk2=2;sk2=4;sk2=8" 2>&1 | grep "^\[" | tail -n +2 | cut -c1-135
done
