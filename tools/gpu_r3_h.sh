#!/bin/bash
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
{
echo "== ablations (persistent kernel)"; timeout 600 python tools/sweep.py --workload glm45air_fp8w8a8_prefill_m8192 --reps 5 --cfgs "dbg=1;dbg=9;dbg=17;dbg=25;dbg=33;dbg=65;dbg=89;dbg=3;dbg=67" 2>&1 | grep "^\[" | cut -c1-100
} > gpurun_out/r3_h.log 2>&1
cat gpurun_out/r3_h.log
