#!/bin/bash
# 32-row token tiles (fp8 / 4-bit decode batches with <= 32 rows per expert): parity + A/B against 64-row tiles
set -u
timeout 400 python -m pytest tests/test_gpu_moe.py -m gpu -x -q -k "quantised_tiled or dequant_is_bit_exact or geometries or fp4_golden or fp8_w8a8" 2>&1 | tail -3
sw() { echo "== $1 M=$2"; timeout 200 python tools/sweep.py --workload $1 --M $2 --reps 20 --cfgs "$3" 2>&1 | grep "^\[" | tail -n +2 | cut -c1-200; }
sw dsv3_ep8_rank_fp8w8a8_rows256 256 ";;tiled=64;tiled=32;tiled=64"
sw dsv3_ep8_rank_fp8w8a16_rows256 256 ";;tiled=64;tiled=32"
sw mixtral8x7b_int4g128_decode_m128 32 ";;tiled=64;tiled=32"
sw mixtral8x7b_int4g128_decode_m128 16 ";;tiled=64;tiled=32"
sw mixtral8x7b_mxfp4_decode_m32 32 ";;tiled=64;tiled=32"
sw mixtral8x7b_fp8w8a8_decode_m32 48 ";;tiled=64;tiled=32"
