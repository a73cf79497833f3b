#!/bin/bash
set -u
mkdir -p gpurun_out
echo "== tests skipped"
echo "== mxfp4 m32"; timeout 600 python tools/sweep.py --workload mixtral8x7b_mxfp4_decode_m32 --cfgs ";;nt2=1;nt1=2;tiled=64,pd1=4,pd2=8" 2>&1 | grep "^\[" | cut -c1-200
echo "== mxfp4 m128"; timeout 600 python tools/sweep.py --workload mixtral8x7b_mxfp4_decode_m128 --cfgs ";;pd1=2,pd2=4;pd1=4,pd2=4;pd1=8,pd2=8;waves=8;tiled=-1" 2>&1 | grep "^\[" | cut -c1-200
echo "== nvfp4 m128"; timeout 600 python tools/sweep.py --workload mixtral8x7b_nvfp4_decode_m128 --cfgs ";;pd1=2,pd2=4;waves=8" 2>&1 | grep "^\[" | cut -c1-200
