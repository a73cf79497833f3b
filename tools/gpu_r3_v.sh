#!/bin/bash
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
bash tools/gpu_pmc.sh glm45air_fp8w8a8_prefill_m8192 "" mem 2>&1 | grep "a8w_kernel" | tee gpurun_out/r3_v_pmc_mem.log
