#!/bin/bash
# round 4, session b: steady-loop ablations of gemm_w4x_kernel (library built with -DLKM_W4X_ABLS; dbg = 1 + 16 * ablation mask:
# 1 no barrier, 2 no token staging, 4 no weight loads, 8 no decode, 16 no MFMA)
mkdir -p gpurun_out
export PYTHONPATH=$PWD
CF="pf=5,tiled=64,waves=4,pd1=2,pd2=2,dbg=1"
for w in 4 8; do for a in 0 1 2 4 8 16 24 6 30 22 14; do CF="$CF;pf=5,tiled=64,waves=$w,pd1=2,pd2=2,dbg=$((1 + 16 * a))"; done; done
timeout 600 python tools/sweep.py --workload mixtral8x7b_int4g128_decode_m128 --reps 30 --cfgs "$CF" > gpurun_out/r4b_abl.log 2>&1
grep "^\[" gpurun_out/r4b_abl.log | cut -c1-120
