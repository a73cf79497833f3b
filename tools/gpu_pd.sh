#!/bin/bash
# prefetch-depth experiment for the tiled kernels (development)
set -u
mkdir -p gpurun_out
echo "== tests"; timeout 900 python -m pytest tests/test_gpu_moe.py -m gpu -q -x --timeout 600 -k "prefetch or tiled or geometries" 2>&1 | tail -3
C='pd1=2,pd2=2;pd1=4,pd2=4;pd1=8,pd2=8;pd1=4,pd2=8;pd1=4,pd2=4,waves=8;pd1=8,pd2=8,waves=8'
echo "== int4 m128"; timeout 600 python tools/sweep.py --workload mixtral8x7b_int4g128_decode_m128 --cfgs "$C" 2>&1 | grep "^\[" | tee gpurun_out/pd_int4.log
echo "== bf16 m128"; timeout 600 python tools/sweep.py --workload mixtral8x7b_bf16_decode_m32 --M 128 --cfgs "$C" 2>&1 | grep "^\[" | tee gpurun_out/pd_bf16_m128.log
echo "== fp8a8 m128"; timeout 600 python tools/sweep.py --workload mixtral8x7b_fp8w8a8_decode_m32 --M 128 --cfgs "$C" 2>&1 | grep "^\[" | tee gpurun_out/pd_fp8a8_m128.log
echo "== dsv3 slice fp8a8"; timeout 600 python tools/sweep.py --workload dsv3_ep8_rank_fp8w8a8_rows256 --cfgs "pd1=2,pd2=2;pd1=4,pd2=4;pd1=8,pd2=8" 2>&1 | grep "^\[" | tee gpurun_out/pd_dsv3.log
echo "== glm bf16 prefill"; timeout 600 python tools/sweep.py --workload glm45air_bf16_prefill_m8192 --reps 5 --cfgs "pd1=2,pd2=2;tiled=128,pd1=4,pd2=4;tiled=128,pd1=2,pd2=2;pd1=2,pd2=4" 2>&1 | grep "^\[" | tee gpurun_out/pd_glm.log
