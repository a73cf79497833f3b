#!/bin/bash
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== pytest gpu"; timeout 2000 python -m pytest tests -m gpu -q --timeout 900 2>&1 | tail -8 | tee gpurun_out/pytest_gpu.log
echo "== GLM fp8-W8A8 prefill"
timeout 600 python tools/sweep.py --workload glm45air_fp8w8a8_prefill_m8192 --reps 10 --cfgs ";xcd=-1;tiled=128" 2>&1 | grep -v "^{" | cut -c1-170 | tee gpurun_out/glm_a8_sweep.log
echo "== bench lines"
for w in glm45air_fp8w8a8_prefill_m8192 glm45air_bf16_prefill_m8192; do
timeout 600 python bench.py --workload $w --steps 20 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | grep '^{' | tee gpurun_out/bench_$w.json | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print(j['config']['workload'], j['ms_per_step'], 'ms', j['roofline']['kernel_ms'], j['roofline']['achieved'], j['roofline']['unit'], j['roofline']['frac'])"
done
