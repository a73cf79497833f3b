#!/bin/bash
# round 3, call F: the persistent version of the register-streaming fp8 prefill kernel
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
{
echo "== debug"; timeout 300 python tests/a8w_debug.py
echo "== pytest a8 prefill"; timeout 900 python -m pytest tests/test_gpu_moe.py -m gpu -q -k "prefill_kernel_scaled_mfma" 2>&1 | tail -25
echo "== config4"; timeout 600 python -m pytest tests/test_gpu_fullsize.py -m gpu -x -q -k "config4 and w8a8" 2>&1 | tail -8
echo "== sweep"; timeout 400 python tools/sweep.py --workload glm45air_fp8w8a8_prefill_m8192 --reps 6 --cfgs ";xcd=1;dbg=1;dbg=2;dbg=3;dbg=2,xcd=1;pf=8" 2>&1 | grep -v "^{" | cut -c1-130
} > gpurun_out/r3_f.log 2>&1
tail -c 5000 gpurun_out/r3_f.log
