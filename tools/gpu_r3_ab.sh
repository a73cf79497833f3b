#!/bin/bash
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python tools/sweep.py --workload glm45air_fp8w8a8_prefill_m8192 --reps 2 --cfgs ";dbg=256;dbg=257" 2>&1 | grep -v '^{\|amdgpu.ids\|^#' | cut -c1-130 | sort | uniq -c | sort -rn | head -40 | tee gpurun_out/r3_ab_waits.log
