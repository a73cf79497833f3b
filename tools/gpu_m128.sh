#!/bin/bash
set -u
for wl in mixtral8x7b_bf16_decode_m32:128 mixtral8x7b_fp8w8a8_decode_m32:128 mixtral8x7b_int4g128_decode_m128:128 mixtral8x7b_mxfp4_decode_m128:128 mixtral8x7b_bf16_decode_m32:512; do
  w=${wl%%:*}; m=${wl##*:}
  echo "== $w M=$m"; timeout 600 python tools/sweep.py --workload $w --M $m --cfgs ";;" 2>&1 | grep "^\[" | tail -1 | cut -c1-215
done
