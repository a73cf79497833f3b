#!/bin/bash
# EPLB on the GPU: the tests that could not run in round 1 (budget spent), then the device-side timings
set -u
mkdir -p gpurun_out
echo "== pytest gpu (shared experts, layer, eplb)"; timeout 900 python -m pytest tests/test_zz1_gpu_shared_experts.py tests/test_zz2_gpu_layer.py tests/test_zz3_gpu_eplb.py -m gpu -q --timeout 300 2>&1 | tail -15
echo "== timings"; timeout 600 python tools/eplb_timing.py 2>&1 | tee gpurun_out/eplb_timing.log | tail -20
