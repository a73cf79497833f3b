#!/bin/bash
# round 3, call B: ablations of the register-streaming fp8 prefill kernel
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
{
echo "== sweep"; timeout 600 python tools/sweep.py --workload glm45air_fp8w8a8_prefill_m8192 --reps 5 --cfgs ";xcd=1;dbg=1;dbg=9;dbg=17;dbg=25;dbg=33;dbg=65;dbg=89;dbg=2;dbg=66;dbg=2,xcd=1;dbg=1,xcd=1" 2>&1 | grep -v "^{" | cut -c1-120
} > gpurun_out/r3_b.log 2>&1
cat gpurun_out/r3_b.log
