#!/bin/bash
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
echo "== pytest"; timeout 1800 python -m pytest tests/test_gpu_moe.py tests/test_gpu_fused_step.py -m gpu -q -x --timeout 900 2>&1 | tail -3
for i in 1 2 3; do
for t in "" "tiled2=64"; do
for r in uniform zipf; do
timeout 300 python bench.py --no-extras --no-cpu-baseline --steps 200 --warmup 20 --routing $r ${t:+--tune $t} 2>/dev/null | grep '^{' | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print('$i tune=[$t] $r step us', round(j['ms_per_step']*1e3,1), j['roofline']['kernel_ms'])"
done; done; done | tee gpurun_out/r3_al.log
