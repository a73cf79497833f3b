#!/bin/bash
# round 4, session a: parity of the 32x32-MFMA 4-bit decode kernels (gemm_w4x.h), then the variant sweep on configs[2]
set -x
mkdir -p gpurun_out
export PYTHONPATH=$PWD
timeout 900 python -m pytest tests/test_gpu_w4x.py -x -q 2>&1 | tail -15 > gpurun_out/r4a_tests.log
cat gpurun_out/r4a_tests.log
CF=""
for t in 32 64; do for w in 4 8; do for pd in 2 4; do for d in 0 1; do CF="$CF;pf=5,tiled=$t,waves=$w,pd1=$pd,pd2=$pd,dbg=$d"; done; done; done; done
timeout 600 python tools/sweep.py --workload mixtral8x7b_int4g128_decode_m128 --reps 30 --cfgs "$CF;tiled=64;tiled=32" > gpurun_out/r4a_sweep.log 2>&1
grep -v "^{" gpurun_out/r4a_sweep.log | tail -40
