#!/bin/bash
set -u
export TMPDIR=/tmp
echo "== pytest gpu"; timeout 1500 python -m pytest tests -m gpu -q --timeout 600 2>&1 | tail -6
for LIB in lvllm_amd/liblkm.so lvllm_amd/liblkm_noslp.so; do
export LKM_LIB_PATH=$PWD/$LIB; echo "#### $LIB"
echo "== bf16 m32"; timeout 600 python tools/sweep.py --M 32 --cfgs ";" 2>&1 | grep -v '^{' | tail -1
echo "== bf16 m128"; timeout 600 python tools/sweep.py --M 128 --cfgs "tiled=-1;tiled=64,waves=8;tiled=64,waves=4" 2>&1 | grep -v '^{' | tail -3
echo "== int4 m32"; timeout 600 python tools/sweep.py --workload mixtral8x7b_int4g128_decode_m128 --M 32 --cfgs ";nt1=2,nt2=2" 2>&1 | grep -v '^{' | tail -2
echo "== int4 m128"; timeout 600 python tools/sweep.py --workload mixtral8x7b_int4g128_decode_m128 --cfgs "tiled=-1;tiled=64,waves=8;tiled=64,waves=4" 2>&1 | grep -v '^{' | tail -3
done
