#!/bin/bash
# round 4, session e: gemm_w4e.h with 4 / 7 / 8 / 14 consumer waves per workgroup on configs[2] (and parity of those variants)
mkdir -p gpurun_out
export PYTHONPATH=$PWD
timeout 900 python -m pytest tests/test_gpu_w4x.py -x -q 2>&1 | tail -8 > gpurun_out/r4e_tests.log
cat gpurun_out/r4e_tests.log
CF="tiled=64;pf=6,tiled=64,pd1=3,pd2=3,dbg=1"
for w in 4 7 8 14; do CF="$CF;pf=6,tiled=64,waves=$w,pd1=3,pd2=3,dbg=1"; done
for w in 4 7 8 14; do CF="$CF;pf=6,tiled=32,waves=$w,pd1=3,pd2=3,dbg=1"; done
timeout 600 python tools/sweep.py --workload mixtral8x7b_int4g128_decode_m128 --reps 30 --cfgs "$CF" > gpurun_out/r4e_sweep.log 2>&1
grep "^\[" gpurun_out/r4e_sweep.log | cut -c1-150
timeout 600 python tools/sweep.py --workload mixtral8x7b_int4g128_decode_m128 --M 64 --reps 30 --cfgs "$CF" > gpurun_out/r4e_sweep64.log 2>&1
grep "^\[" gpurun_out/r4e_sweep64.log | cut -c1-150
