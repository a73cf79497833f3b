#!/bin/bash
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
for i in 1 2; do
for t in "" "pd1=4,pd2=4" "tiled=32" "waves=8" "pd1=4"; do
timeout 300 python bench.py --workload dsv3_fp8w8a8_ep_decode_b256 --no-extras --no-cpu-baseline --steps 100 --warmup 10 ${t:+--tune $t} 2>/dev/null | grep '^{' | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print('$i dsv3 tune=[$t] step us', round(j['ms_per_step']*1e3,1), j['roofline']['kernel_ms'])"
done
for t in "" "pd1=4,pd2=4" "pd2=4" "pd1=4"; do
timeout 300 python bench.py --workload mixtral8x7b_int4g128_decode_m128 --no-extras --no-cpu-baseline --steps 200 --warmup 20 ${t:+--tune $t} 2>/dev/null | grep '^{' | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print('$i int4 tune=[$t] step us', round(j['ms_per_step']*1e3,1), j['roofline']['kernel_ms'])"
done
done | tee gpurun_out/r3_am.log
