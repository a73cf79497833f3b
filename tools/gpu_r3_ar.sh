#!/bin/bash
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
for i in 1 2 3; do
for w in 5; do
for nr in 0 1; do
LKM_BENCH_NO_PREROLL=$nr timeout 300 python bench.py --no-extras --no-cpu-baseline --steps 20 --warmup $w 2>/dev/null | grep '^{' | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print('$i warmup=$w no_rewarm=$nr step us', round(j['ms_per_step']*1e3,1), 'long', j.get('long_run',{}).get('ms_per_step'))"
done; done; done | tee gpurun_out/r3_ar.log
for nr in 0 1; do
LKM_BENCH_NO_PREROLL=$nr timeout 300 python bench.py --workload mixtral8x7b_int4g128_decode_m128 --no-extras --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | grep '^{' | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print('int4 no_rewarm=$nr step us', round(j['ms_per_step']*1e3,1), 'long', j.get('long_run',{}).get('ms_per_step'))"
done | tee -a gpurun_out/r3_ar.log
