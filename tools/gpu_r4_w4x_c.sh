#!/bin/bash
# round 4, session c: parity of the loader / consumer kernel (gemm_w4e.h, pf = 6) and its variant sweep on configs[2]
mkdir -p gpurun_out
export PYTHONPATH=$PWD
timeout 900 python -m pytest tests/test_gpu_w4x.py -x -q 2>&1 | tail -15 > gpurun_out/r4c_tests.log
cat gpurun_out/r4c_tests.log
CF="tiled=64;pf=5,tiled=64,waves=4,pd1=2,pd2=2,dbg=1"
for t in 64 32; do for pd in 3 4; do for d in 0 1; do CF="$CF;pf=6,tiled=$t,pd1=$pd,pd2=$pd,dbg=$d"; done; done; done
timeout 600 python tools/sweep.py --workload mixtral8x7b_int4g128_decode_m128 --reps 30 --cfgs "$CF" > gpurun_out/r4c_sweep.log 2>&1
grep "^\[" gpurun_out/r4c_sweep.log | cut -c1-330
