#!/bin/bash
# Which decode plan is robust to routing skew?  The planner's thresholds were measured under uniform routing (rounds 1-3); the
# host cannot see the skew of a step.  Per-kernel HIP events of eager steps (tools/sweep.py), uniform and Zipf, default plan
# against the alternatives, at the batch sizes around each threshold.   bash tools/plan_robustness_sweep.sh   (GPU box)
run() {  # workload M cfgs
  for r in uniform zipf; do
    echo "== $1 M=$2 $r"
    python tools/sweep.py --workload $1 --M $2 --routing $r --reps 30 --cfgs "$3" 2>&1 | grep "^\[" | cut -c1-170
  done
}
for m in 16 24 32 40; do run mixtral8x7b_bf16_decode_m32 $m ";tiled=64;tiled=32;tiled2=-1,tiled=-1"; done
for m in 16 32 40 48 64; do run mixtral8x7b_fp8w8a8_decode_m32 $m ";tiled=64;tiled=32;tiled=-1"; done
for m in 32 64 128 256; do run mixtral8x7b_int4g128_decode_m128 $m ";tiled=64;tiled=32;pf=5,tiled=64;pf=6,tiled=64"; done
for m in 32 64 128; do run dsv3_ep8_rank_fp8w8a8_rows256 $m ";tiled=64;tiled=32;tiled=-1"; done
