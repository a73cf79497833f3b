#!/bin/bash
# round 4, session d: parity of gemm_w4e.h, then SQ counter passes of the three 4-bit GEMM1 structures on configs[2]
mkdir -p gpurun_out
export PYTHONPATH=$PWD
timeout 900 python -m pytest tests/test_gpu_w4x.py -x -q 2>&1 | tail -8 > gpurun_out/r4d_tests.log
cat gpurun_out/r4d_tests.log
W=mixtral8x7b_int4g128_decode_m128
for cfg in "tiled=64" "pf=5,tiled=64,waves=4,pd1=2,pd2=2,dbg=1" "pf=6,tiled=64,pd1=3,pd2=3,dbg=1"; do
  echo "=== $cfg"
  bash tools/gpu_pmc.sh $W "$cfg" sq 2>&1 | grep -v "^$" | grep "gemm_tiled_kernel<3, 1, 1, 4\|w4x_kernel\|w4e_kernel"
done > gpurun_out/r4d_pmc.log 2>&1
cat gpurun_out/r4d_pmc.log | cut -c1-200
