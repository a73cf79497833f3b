#!/bin/bash
# round 3, call M: mixed plan (streamer GEMM1 + tile GEMM2), capacity test again
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
echo "== pytest"; timeout 1500 python -m pytest tests/test_zz5_gpu_create_near_capacity.py tests/test_gpu_fused_step.py tests/test_gpu_moe.py -m gpu -q -x --timeout 900 2>&1 | tail -15 | tee gpurun_out/r3_m_pytest.log
for r in uniform zipf; do
for M in 32 24 20; do
echo "== mixtral bf16 M=$M $r"
timeout 300 python tools/sweep.py --workload mixtral8x7b_bf16_decode_m32 --M $M --routing $r --cfgs ";tiled2=-1;tiled2=32" 2>&1 | grep -v '^{' | tee -a gpurun_out/r3_m_sweep.log
done
echo "== mixtral fp8w8a8 M=32 $r"
timeout 300 python tools/sweep.py --workload mixtral8x7b_fp8w8a8_decode_m32 --routing $r --cfgs ";tiled2=64;tiled2=32" 2>&1 | grep -v '^{' | tee -a gpurun_out/r3_m_sweep.log
echo "== dsv3 slice fp8w8a8 $r"
timeout 300 python tools/sweep.py --workload dsv3_ep8_rank_fp8w8a8_rows256 --routing $r --cfgs ";tiled=32,pd1=4;tiled=32,pd1=4,pd2=4;tiled=32,waves=8" 2>&1 | grep -v '^{' | tee -a gpurun_out/r3_m_sweep.log
done
