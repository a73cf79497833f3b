#!/bin/bash
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
echo "== pytest"; timeout 1800 python -m pytest tests/test_gpu_moe.py tests/test_gpu_fused_step.py tests/test_gpu_fullsize.py tests/test_gpu_routing.py -m gpu -q -x --timeout 900 2>&1 | tail -4
timeout 900 python tools/sweep.py --workload glm45air_fp8w8a8_prefill_m8192 --reps 5 --cfgs ";;" 2>&1 | grep -v '^{\|amdgpu.ids\|^#' | cut -c1-110 | tee gpurun_out/r3_y.log
timeout 900 python tools/sweep.py --workload mixtral8x7b_bf16_decode_m32 --reps 20 --cfgs ";;" 2>&1 | grep -v '^{\|amdgpu.ids\|^#' | cut -c1-110 | tee -a gpurun_out/r3_y.log
timeout 900 python tools/sweep.py --workload glm45air_bf16_prefill_m8192 --reps 3 --cfgs ";" 2>&1 | grep -v '^{\|amdgpu.ids\|^#' | cut -c1-110 | tee -a gpurun_out/r3_y.log
