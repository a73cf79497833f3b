#!/bin/bash
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
bash tools/gpu_pmc.sh mixtral8x7b_int4g128_decode_m128 "" sq 2>&1 | grep "gemm_tiled_kernel" | tee gpurun_out/r03_int4_pmc_raw.log
