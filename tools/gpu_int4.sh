#!/bin/bash
# int4 decode: parity tests of the decoder + geometry sweep at the BASELINE config (M=128) and M=32/512
set -u
timeout 600 python -m pytest tests/test_gpu_moe.py -m gpu -x -q -k "int4 or quantised or randomised or prefetch or geometries" 2>&1 | tail -3
W=mixtral8x7b_int4g128_decode_m128
echo "== int4 M=128"; timeout 300 python tools/sweep.py --workload $W --reps 20 --cfgs ";;waves=8;pd1=4;waves=8,pd1=4;pd2=2;tiled=128,waves=8;tiled=-1" 2>&1 | grep "^\[" | cut -c1-230
echo "== int4 M=32"; timeout 300 python tools/sweep.py --workload $W --M 32 --reps 20 --cfgs ";;nt1=1;nt1=2;tiled=64" 2>&1 | grep "^\[" | cut -c1-230
echo "== int4 M=512"; timeout 300 python tools/sweep.py --workload $W --M 512 --reps 10 --cfgs ";;tiled=64;tiled=128,waves=8" 2>&1 | grep "^\[" | cut -c1-230
