#!/bin/bash
# round 3, call P: int4 decode landscape at Mixtral M=128 (configs[2]) + ep test
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
echo "== pytest ep"; timeout 900 python -m pytest tests/test_gpu_ep.py -m gpu -q --timeout 900 2>&1 | tail -5
for w in mixtral8x7b_int4g128_decode_m128 mixtral8x7b_int4g128_fast_decode_m128; do
echo "== $w"
timeout 600 python tools/sweep.py --workload $w --cfgs ";pd1=4;pd1=4,pd2=4;tiled=32;tiled=32,pd1=4;nt1=2;waves=8;pf=4;tiled=128" 2>&1 | grep -v '^{\|amdgpu.ids' | cut -c1-330 | tee -a gpurun_out/r3_p_sweep.log
done
