#!/bin/bash
# round 3, call L: streaming creation tests, heavy-first launch order under Zipf routing
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
echo "== pytest create / sort / skew"; timeout 1500 python -m pytest tests/test_gpu_create_streaming.py tests/test_zz5_gpu_create_near_capacity.py tests/test_gpu_routing.py tests/test_gpu_router.py "tests/test_gpu_moe.py::test_hybrid_dispatch_skewed_routing" tests/test_gpu_fused_step.py -m gpu -q -x --timeout 900 2>&1 | tail -15 | tee gpurun_out/r3_l_pytest.log
for r in uniform zipf; do
echo "== mixtral bf16 M=32 $r"
timeout 300 python tools/sweep.py --workload mixtral8x7b_bf16_decode_m32 --routing $r --cfgs ";tbmax=1;tbmax=4;tiled=64;tiled=32" 2>&1 | grep -v '^{' | tee -a gpurun_out/r3_l_sweep.log
echo "== dsv3 slice fp8w8a8 $r"
timeout 300 python tools/sweep.py --workload dsv3_ep8_rank_fp8w8a8_rows256 --routing $r --cfgs ";tiled=-1;tiled=-1,kw1=2;tiled=-1,kw1=4;pd1=4,pd2=4;waves=8;tiled=32;tiled=128" 2>&1 | grep -v '^{' | tee -a gpurun_out/r3_l_sweep.log
done
