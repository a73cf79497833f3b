#!/bin/bash
# round 2, GPU session C: the fp8 x fp8 prefill kernel on the scaled MFMA -- parity, then GLM-4.5-Air prefill timings
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== parity"; timeout 900 python -m pytest tests/test_gpu_moe.py tests/test_gpu_fullsize.py -m gpu -q -x --timeout 600 -k "prefill_kernel_scaled or config4 or w8a8" 2>&1 | tail -15
echo "== GLM fp8-W8A8 prefill: new kernel (auto), no XCD runs, legacy 128-row tiled kernel"
timeout 600 python tools/sweep.py --workload glm45air_fp8w8a8_prefill_m8192 --reps 10 --cfgs ";xcd=-1;tiled=128;tiled=128,xcd=1" 2>&1 | grep -v "^{" | tee gpurun_out/glm_a8_sweep.log
echo "== Mixtral fp8-W8A8 prefill M=4096"
timeout 600 python tools/sweep.py --workload mixtral8x7b_fp8w8a8_decode_m32 --M 4096 --reps 10 --cfgs ";xcd=1;tiled=128" 2>&1 | grep -v "^{" | tee -a gpurun_out/glm_a8_sweep.log
echo "== bench line"; timeout 600 python bench.py --workload glm45air_fp8w8a8_prefill_m8192 --steps 20 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | grep '^{' | tee gpurun_out/glm_a8_bench.json | cut -c1-700
