#!/bin/bash
# 16-bit weights, many experts, few rows per expert (decode): the hybrid plan (streamer + tile list for the experts with
# more than 16*tb rows) against plain 64-row tiles -- per-kernel HIP events of eager steps (tools/sweep.py), uniform and Zipf.
#   bash tools/hybrid_vs_tiles_ab.sh      (on the GPU box)
for spec in "dsv3_ep8_rank_bf16_rows256 256" "dsv3_ep8_rank_bf16_rows256 128" "dsv3_ep8_rank_bf16_rows256 64" "qwen3_30b_a3b_bf16_decode_m1 32" "qwen3_30b_a3b_bf16_decode_m1 64" "qwen3_30b_a3b_bf16_decode_m1 128" "qwen3_30b_a3b_bf16_decode_m1 256" "mixtral8x7b_bf16_decode_m32 48"; do
  set -- $spec
  for r in uniform zipf; do
    echo "== $1 M=$2 $r"
    python tools/sweep.py --workload $1 --M $2 --routing $r --reps 30 --cfgs ";tiled=64;tiled=32;hybrid=-1" 2>&1 | grep "^\[" | cut -c1-150
  done
done
