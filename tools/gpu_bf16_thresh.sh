#!/bin/bash
set -u
timeout 600 python -m pytest tests/test_gpu_moe.py -m gpu -x -q 2>&1 | tail -2
for m in 48 64 96; do
  echo "== bf16 M=$m"
  timeout 300 python tools/sweep.py --workload mixtral8x7b_bf16_decode_m32 --M $m --reps 12 --cfgs ";;tiled=64;tiled=64,pd1=2" 2>&1 | grep "^\[" | tail -n +2 | cut -c1-135
  echo "== fp8a8 M=$m"
  timeout 300 python tools/sweep.py --workload mixtral8x7b_fp8w8a8_decode_m32 --M $m --reps 12 --cfgs ";;tiled=64;tiled=-1" 2>&1 | grep "^\[" | tail -n +2 | cut -c1-135
done
echo "== dsv3 bf16"; timeout 300 python tools/sweep.py --workload dsv3_ep8_rank_bf16_rows256 --reps 12 --cfgs ";;tiled=64;hybrid=-1" 2>&1 | grep "^\[" | tail -n +2 | cut -c1-135
echo "== dsv3 fp8a8 (auto now tiled)"; timeout 300 python tools/sweep.py --workload dsv3_ep8_rank_fp8w8a8_rows256 --reps 12 --cfgs ";;waves=8;pd1=4;tiled=128,waves=8" 2>&1 | grep "^\[" | tail -n +2 | cut -c1-135
