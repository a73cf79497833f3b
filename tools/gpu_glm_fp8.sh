#!/bin/bash
set -u
timeout 600 python -m pytest tests/test_gpu_moe.py -m gpu -x -q -k "fp8 or prefetch or geometries or randomised" 2>&1 | tail -2
echo "== glm fp8a8"; timeout 300 python tools/sweep.py --workload glm45air_fp8w8a8_prefill_m8192 --reps 6 --cfgs ";;tiled=128,waves=8;tiled=128,waves=8,nt2=2;tiled=128,waves=8,xcd=1;tiled=64,waves=8" 2>&1 | grep "^\[" | tail -n +2 | cut -c1-135
echo "== mixtral fp8a8 M=512/2048"; for m in 512 2048; do timeout 300 python tools/sweep.py --workload mixtral8x7b_fp8w8a8_decode_m32 --M $m --reps 6 --cfgs ";;tiled=128,waves=8" 2>&1 | grep "^\[" | tail -n +2 | cut -c1-135; done
