#!/bin/bash
set -u
mkdir -p gpurun_out
echo "== tests"; timeout 1200 python -m pytest tests -m gpu -q -x --timeout 600 2>&1 | tail -3
for wl in mixtral8x7b_bf16_decode_m32:128 mixtral8x7b_fp8w8a8_decode_m32:128 mixtral8x7b_int4g128_decode_m128:128 mixtral8x7b_mxfp4_decode_m128:128 mixtral8x7b_mxfp4_decode_m32:32 mixtral8x7b_int4g128_decode_m128:32 mixtral8x7b_bf16_decode_m32:256; do
  w=${wl%%:*}; m=${wl##*:}
  echo "== $w M=$m"; timeout 600 python tools/sweep.py --workload $w --M $m --cfgs ";;waves=8;pd1=2,pd2=4" 2>&1 | grep "^\[" | tail -3 | cut -c1-215
done
