#!/bin/bash
set -u
export TMPDIR=/tmp
echo "== pytest gpu"; timeout 1500 python -m pytest tests -m gpu -q --timeout 600 2>&1 | tail -5
echo "== int4 m128"; timeout 600 python tools/sweep.py --workload mixtral8x7b_int4g128_decode_m128 --cfgs ";tiled=64,waves=8;hybrid=-1,tiled=-1" 2>&1 | grep -v '^{' | tail -3
echo "== bf16 m128"; timeout 600 python tools/sweep.py --M 128 --cfgs ";tiled=64,waves=8" 2>&1 | grep -v '^{' | tail -2
echo "== bf16 m512"; timeout 600 python tools/sweep.py --M 512 --cfgs ";tiled=128,waves=8" 2>&1 | grep -v '^{' | tail -2
echo "== glm"; timeout 600 python tools/sweep.py --workload glm45air_bf16_prefill_m8192 --reps 5 --cfgs ";" 2>&1 | grep -v '^{' | tail -1
