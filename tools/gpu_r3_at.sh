#!/bin/bash
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
for i in 1 2; do
for w in mixtral8x7b_int4g128_decode_m128 mixtral8x7b_int4g128_fast_decode_m128 mixtral8x7b_mxfp4_decode_m128 mixtral8x7b_nvfp4_decode_m128; do
for t in "" "waves=8" "tiled=32" "pf=4"; do
timeout 300 python bench.py --workload $w --no-extras --no-cpu-baseline --steps 100 --warmup 10 ${t:+--tune $t} 2>/dev/null | grep '^{' | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print('$i $w tune=[$t] step us', round(j['ms_per_step']*1e3,1), j['roofline']['kernel_ms'])"
done; done; done | sed 's/mixtral8x7b_//' | tee gpurun_out/r3_at.log
