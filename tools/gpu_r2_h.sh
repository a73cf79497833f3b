#!/bin/bash
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -f gpurun_out/pmc_int4.log
for w in mixtral8x7b_int4g128_decode_m128 mixtral8x7b_int4g128_fast_decode_m128 mixtral8x7b_mxfp4_decode_m128; do
  echo "=========== PMC $w" | tee -a gpurun_out/pmc_int4.log
  bash tools/gpu_pmc.sh $w "" all 2>&1 | grep -i "gemm_tiled" | tee -a gpurun_out/pmc_int4.log
done
