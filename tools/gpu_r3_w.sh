#!/bin/bash
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
echo "== pytest a8"; timeout 1500 python -m pytest tests/test_gpu_moe.py tests/test_gpu_fullsize.py -m gpu -q -x --timeout 900 -k "w8a8 or a8 or prefill or fp8 or config4 or glm" 2>&1 | tail -4
timeout 900 python tools/sweep.py --workload glm45air_fp8w8a8_prefill_m8192 --reps 5 --cfgs ";;dbg=2;dbg=1;dbg=17;dbg=25" 2>&1 | grep -v '^{\|amdgpu.ids\|^#' | cut -c1-110 | tee gpurun_out/r3_w.log
