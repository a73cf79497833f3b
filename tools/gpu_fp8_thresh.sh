#!/bin/bash
# fp8 formats: skinny streamer vs 64-row tiles at decode batches; plus the full GPU suite after a planner change
set -u
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -2
for wl in mixtral8x7b_fp8w8a8_decode_m32; do
for m in 16 32 64 96; do
  echo "== $wl M=$m"
  timeout 300 python tools/sweep.py --workload $wl --M $m --reps 20 --cfgs ";;tiled=64;tiled=64,pd1=4" 2>&1 | grep "^\[" | tail -n +2 | cut -c1-135
done
done
for wl in dsv3_ep8_rank_fp8w8a8_rows256 dsv3_ep8_rank_fp8w8a16_rows256; do
  echo "== $wl"
  timeout 300 python tools/sweep.py --workload $wl --reps 20 --cfgs ";;tiled=64;hybrid=-1" 2>&1 | grep "^\[" | tail -n +2 | cut -c1-135
done
