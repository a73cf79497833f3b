#!/bin/bash
set -u
for m in 192 256 384 512 768 1024 1536; do
echo "== bf16 M=$m"; timeout 600 python tools/sweep.py --workload mixtral8x7b_bf16_decode_m32 --M $m --reps 8 --cfgs ";tiled=64;tiled=128,waves=8;tiled=256,waves=8,nt2=2" 2>&1 | grep "^\[" | tail -3 | cut -c1-110
done
for m in 256 512 1024; do
echo "== fp8a8 M=$m"; timeout 600 python tools/sweep.py --workload mixtral8x7b_fp8w8a8_decode_m32 --M $m --reps 8 --cfgs ";tiled=64;tiled=128,waves=8" 2>&1 | grep "^\[" | tail -2 | cut -c1-110
echo "== mxfp4 M=$m"; timeout 600 python tools/sweep.py --workload mixtral8x7b_mxfp4_decode_m32 --M $m --reps 8 --cfgs ";tiled=64;tiled=128,waves=8" 2>&1 | grep "^\[" | tail -2 | cut -c1-110
done
