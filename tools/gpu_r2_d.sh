#!/bin/bash
# round 2, GPU session D: parity of the fp8 prefill kernel + PMC passes on GLM-4.5-Air fp8 prefill, with and without XCD runs
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== parity"; timeout 900 python -m pytest tests/test_gpu_moe.py tests/test_gpu_fullsize.py -m gpu -q --timeout 600 -k "prefill_kernel_scaled or config4 or w8a8" 2>&1 | tail -5
for cfg in "xcd=-1" ""; do
  echo "=========== PMC cfg=[$cfg]"
  bash tools/gpu_pmc.sh glm45air_fp8w8a8_prefill_m8192 "$cfg" all 2>&1 | grep -i "prefill_a8\|gemm" | tee -a gpurun_out/pmc_glm_a8.log
done
