#!/bin/bash
# DSv3 EP=8 rank slice (32 experts, 256 rows, fp8 W8A8): knob sweep under Zipf and uniform routing
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
L=gpurun_out/r3_slice_sweep.log; : > $L
W=dsv3_ep8_rank_fp8w8a8_rows256
for r in zipf uniform; do
for t in "" "waves=8" "pd1=2" "pd1=8" "tiled=64" "tiled=64,waves=8" "xcd=1" "nt1=2" "tiled=-1" "tiled=32,pd1=4,pd2=2"; do
timeout 200 python bench.py --workload $W --routing $r --no-extras --no-cpu-baseline --steps 200 --warmup 20 ${t:+--tune $t} 2>/dev/null | grep '^{' | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print('%-8s tune=[%-20s] step us %7.1f  %s frac %s' % ('$r', '$t', j['ms_per_step']*1e3, j['roofline']['kernel_ms'], j['roofline']['frac']))" >> $L
done; done
cat $L
