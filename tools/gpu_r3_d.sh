#!/bin/bash
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
{
echo "== tile-major"; timeout 600 python tools/sweep.py --workload glm45air_fp8w8a8_prefill_m8192 --reps 5 --cfgs ";xcd=1;dbg=2;dbg=2,xcd=1" 2>&1 | grep "^\[" | cut -c1-110
echo "== unit-major"; LKM_W_UNIT_MAJOR=1 timeout 600 python tools/sweep.py --workload glm45air_fp8w8a8_prefill_m8192 --reps 5 --cfgs ";xcd=1;dbg=2;dbg=2,xcd=1" 2>&1 | grep "^\[" | cut -c1-110
} > gpurun_out/r3_d.log 2>&1
cat gpurun_out/r3_d.log
bash tools/gpu_pmc.sh glm45air_fp8w8a8_prefill_m8192 "" mem 2>&1 | tail -12
bash tools/gpu_pmc.sh glm45air_fp8w8a8_prefill_m8192 "xcd=1" mem 2>&1 | tail -12
