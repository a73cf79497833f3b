#!/bin/bash
# EPLB on the GPU: the tests that could not run in round 1 (budget spent), then the device-side timings
set -u
mkdir -p gpurun_out
echo "== pytest gpu (shared experts, layer, eplb)"; timeout 900 python -m pytest tests/test_zz1_gpu_shared_experts.py tests/test_zz2_gpu_layer.py tests/test_zz3_gpu_eplb.py -m gpu -q --timeout 300 2>&1 | tail -15
echo "== timings"; timeout 600 python tools/eplb_timing.py 2>&1 | tee gpurun_out/eplb_timing.log | tail -20
echo "== cold-cache variant (SURVEY 8d): headline + the 75 MB Qwen3 layer, where the Infinity Cache matters"
for w in mixtral8x7b_bf16_decode_m32 qwen3_30b_a3b_bf16_decode_m1; do
  timeout 600 python bench.py --workload $w --flush-cache --no-cpu-baseline 2>/dev/null | grep '^{' | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print(j['config']['workload'], 'warm', j['ms_per_step'], 'ms  cold', j.get('ms_per_step_cold'), 'ms')" | tee -a gpurun_out/eplb_timing.log
done
