#!/bin/bash
# round 3, call N: plan rules after the Zipf sweeps; full GPU moe tests
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
echo "== pytest"; timeout 2400 python -m pytest tests/test_zz5_gpu_create_near_capacity.py tests/test_gpu_fused_step.py tests/test_gpu_moe.py tests/test_gpu_ep_rank_shapes.py tests/test_gpu_fullsize.py -m gpu -q -x --timeout 900 2>&1 | tail -15 | tee gpurun_out/r3_n_pytest.log
for r in uniform zipf; do
for w in dsv3_ep8_rank_fp8w8a8_rows256 dsv3_ep8_rank_fp8w8a16_rows256; do
echo "== $w $r"
timeout 300 python tools/sweep.py --workload $w --routing $r --cfgs ";tiled=64;tiled=64,pd1=4,pd2=4" 2>&1 | grep -v '^{\|amdgpu.ids' | cut -c1-260 | tee -a gpurun_out/r3_n_sweep.log
done
echo "== mixtral fp8w8a16 M=32 $r"
timeout 300 python tools/sweep.py --workload mixtral8x7b_fp8w8a8_decode_m32 --M 40 --routing $r --cfgs ";tiled2=-1" 2>&1 | grep -v '^{\|amdgpu.ids' | cut -c1-260 | tee -a gpurun_out/r3_n_sweep.log
done
