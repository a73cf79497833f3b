#!/bin/bash
set -u
timeout 600 python -m pytest tests/test_gpu_moe.py -m gpu -x -q -k "prefill or geometries or randomised" 2>&1 | tail -2
for wl in glm45air_bf16_prefill_m8192; do
  echo "== $wl"; timeout 300 python tools/sweep.py --workload $wl --reps 6 --cfgs ";;xcd=-1;xcd=1" 2>&1 | grep "^\[" | tail -n +2 | cut -c1-135
done
echo "== mixtral bf16 M=1024 / 4096"; for m in 1024 4096; do timeout 300 python tools/sweep.py --workload mixtral8x7b_bf16_decode_m32 --M $m --reps 6 --cfgs ";;xcd=-1" 2>&1 | grep "^\[" | tail -n +2 | cut -c1-135; done
