#!/bin/bash
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
for r in uniform zipf uniform zipf; do
timeout 300 python tools/sweep.py --workload mixtral8x7b_bf16_decode_m32 --routing $r --reps 30 --cfgs ";kw1=2;nt1=2;tbmax=4" 2>&1 | grep -v '^{\|amdgpu.ids\|^# rows' | cut -c1-150 | tee -a gpurun_out/r3_ah.log
done
