#!/bin/bash
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
bash tools/gpu_pmc.sh glm45air_fp8w8a8_prefill_m8192 "xcd=1" sq 2>&1 | grep "a8w_kernel<1, true, true" | tee gpurun_out/r3_i_pmc_sq_xcd.log
