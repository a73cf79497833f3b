#!/bin/bash
set -u
echo "== test"; timeout 900 python -m pytest tests/test_gpu_moe.py -m gpu -q -x --timeout 600 -k "geometries or multi_tile or prefetch or quantised or prefill" 2>&1 | tail -2
for wl in glm45air_bf16_prefill_m8192 glm45air_fp8w8a8_prefill_m8192; do
echo "== $wl"; timeout 900 python tools/sweep.py --workload $wl --reps 5 --cfgs ";;" 2>&1 | grep "^\[" | tail -1 | cut -c1-160
done
echo "== bf16 m512"; timeout 600 python tools/sweep.py --workload mixtral8x7b_bf16_decode_m32 --M 512 --cfgs ";;tiled=128,waves=8" 2>&1 | grep "^\[" | tail -2 | cut -c1-160
echo "== bf16 m2048"; timeout 600 python tools/sweep.py --workload mixtral8x7b_bf16_decode_m32 --M 2048 --reps 5 --cfgs ";;tiled=128,waves=8;tiled=64" 2>&1 | grep "^\[" | tail -3 | cut -c1-160
