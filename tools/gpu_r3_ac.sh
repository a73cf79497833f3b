#!/bin/bash
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python tools/sweep.py --workload glm45air_fp8w8a8_prefill_m8192 --reps 5 --cfgs ";;dbg=512;;dbg=512;;dbg=512" 2>&1 | grep -v '^{\|amdgpu.ids\|^#\|^a8w' | cut -c1-110 | tee gpurun_out/r3_ac.log
