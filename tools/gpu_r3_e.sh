#!/bin/bash
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
{
echo "== overhead ablations"; timeout 600 python tools/sweep.py --workload glm45air_fp8w8a8_prefill_m8192 --reps 5 --cfgs "dbg=3;dbg=67;dbg=1;dbg=2,xcd=1" 2>&1 | grep "^\[" | cut -c1-110
} > gpurun_out/r3_e.log 2>&1
cat gpurun_out/r3_e.log
