#!/bin/bash
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
for i in 1 2 3; do
for w in 5 500 3000; do
timeout 300 python bench.py --no-extras --no-cpu-baseline --steps 20 --warmup $w 2>/dev/null | grep '^{' | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print('$i warmup=$w step us', round(j['ms_per_step']*1e3,1), j['roofline']['kernel_ms'])"
done; done | tee gpurun_out/r3_ap.log
for i in 1 2; do
for s in 20 200 2000; do
timeout 300 python bench.py --no-extras --no-cpu-baseline --steps $s --warmup 5 2>/dev/null | grep '^{' | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print('$i steps=$s step us', round(j['ms_per_step']*1e3,1))"
done; done | tee -a gpurun_out/r3_ap.log
