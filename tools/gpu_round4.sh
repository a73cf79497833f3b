#!/bin/bash
set -u
export TMPDIR=/tmp
python tools/dbg_int4.py 2>&1 | grep -v amdgpu.ids | tail -10 | cut -c1-200
echo "== pytest gpu"; timeout 1500 python -m pytest tests -m gpu -q --timeout 600 2>&1 | tail -8
CF="tiled=-1;tiled=64,waves=8;tiled=64,waves=4"
echo "== int4 m128"; timeout 600 python tools/sweep.py --workload mixtral8x7b_int4g128_decode_m128 --cfgs "$CF" 2>&1 | grep -v '^{' | tail -4
echo "== int4 m32"; timeout 600 python tools/sweep.py --workload mixtral8x7b_int4g128_decode_m128 --M 32 --cfgs ";tiled=64,waves=4" 2>&1 | grep -v '^{' | tail -3
