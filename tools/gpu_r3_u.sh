#!/bin/bash
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
echo "== pytest a8"; timeout 1500 python -m pytest tests/test_gpu_moe.py -m gpu -q -x --timeout 900 -k "w8a8 or a8 or prefill or fp8" 2>&1 | tail -6
timeout 900 python tools/sweep.py --workload glm45air_fp8w8a8_prefill_m8192 --reps 5 --cfgs ";;dbg=2;dbg=1;xcd=-1" 2>&1 | grep -v '^{\|amdgpu.ids\|^#' | cut -c1-110 | tee gpurun_out/r3_u.log
timeout 900 python tools/sweep.py --workload glm45air_fp8w8a8_prefill_m8192 --routing zipf --reps 5 --cfgs ";;pf=8" 2>&1 | grep -v '^{\|amdgpu.ids\|^#' | cut -c1-110 | tee -a gpurun_out/r3_u.log
