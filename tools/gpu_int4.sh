#!/bin/bash
set -u
mkdir -p gpurun_out
echo "== tests"; timeout 900 python -m pytest tests/test_gpu_moe.py -m gpu -q -x --timeout 600 -k "int4 or prefetch or quantised" 2>&1 | tail -2
C=';pd1=2,pd2=2;pd1=2,pd2=4;pd1=4,pd2=4;pd1=8,pd2=8;pd1=4,pd2=8,waves=8;pd1=2,pd2=4,waves=8'
echo "== int4 m128"; timeout 600 python tools/sweep.py --workload mixtral8x7b_int4g128_decode_m128 --cfgs "$C" 2>&1 | grep "^\[" | cut -c1-200
echo "== int4 m32"; timeout 600 python tools/sweep.py --workload mixtral8x7b_int4g128_decode_m128 --M 32 --cfgs ";tiled=64,pd1=4,pd2=8" 2>&1 | grep "^\[" | cut -c1-200
echo "== int4 m512"; timeout 600 python tools/sweep.py --workload mixtral8x7b_int4g128_decode_m128 --M 512 --cfgs ";pd1=2,pd2=2;tiled=128,waves=8,pd1=2,pd2=2;tiled=128,waves=8,pd1=4,pd2=4" 2>&1 | grep "^\[" | cut -c1-200
