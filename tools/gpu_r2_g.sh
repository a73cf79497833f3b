#!/bin/bash
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
for w in mixtral8x7b_int4g128_decode_m128 mixtral8x7b_mxfp4_decode_m128 mixtral8x7b_int4g128_fast_decode_m128; do
timeout 600 python tools/sweep.py --workload $w --reps 20 --cfgs "pf=-1;;nt1=2;nt1=2,pd1=3;nt1=2,tiled=32;nt1=2,nt2=1" 2>&1 | grep -v "^{" | cut -c1-215 | tee -a gpurun_out/w4dma_sweep2.log
done
