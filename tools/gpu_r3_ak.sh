#!/bin/bash
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
for i in 1 2; do
for t in "" "tiled2=32" "pd2=2" "sk2=2" "tiled2=-1" "tiled2=128"; do
for r in uniform zipf; do
timeout 300 python bench.py --no-extras --no-cpu-baseline --steps 200 --warmup 20 --routing $r ${t:+--tune $t} 2>/dev/null | grep '^{' | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print('$i tune=[$t] $r step us', round(j['ms_per_step']*1e3,1), j['roofline']['kernel_ms'])"
done; done; done | tee gpurun_out/r3_ak.log
