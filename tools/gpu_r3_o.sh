#!/bin/bash
# round 3, call O: GPU suite for the plan changes (moe, fused step, ep shapes, full size, routing, create)
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
echo "== pytest"; timeout 2400 python -m pytest tests/test_zz5_gpu_create_near_capacity.py tests/test_gpu_fused_step.py tests/test_gpu_moe.py tests/test_gpu_ep_rank_shapes.py tests/test_gpu_fullsize.py tests/test_gpu_routing.py tests/test_gpu_ep.py -m gpu -q --timeout 900 2>&1 | tail -25 | tee gpurun_out/r3_o_pytest.log
