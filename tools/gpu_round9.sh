#!/bin/bash
set -u
export TMPDIR=/tmp
echo "== pytest gpu"; timeout 1500 python -m pytest tests -m gpu -q --timeout 600 2>&1 | tail -8
echo "== bf16 m32"; timeout 600 python tools/sweep.py --M 32 --cfgs ";tbmax=1;nt1=2" 2>&1 | grep -v '^{' | tail -3
echo "== bf16 m128"; timeout 600 python tools/sweep.py --M 128 --cfgs ";tiled=-1" 2>&1 | grep -v '^{' | tail -2
echo "== mixtral fp8 w8a8 m32"; timeout 600 python tools/sweep.py --workload mixtral8x7b_fp8w8a8_decode_m32 --cfgs ";nt1=2;nt2=2,sk2=2;nt1=2,nt2=2,sk2=2" 2>&1 | grep -v '^{' | tail -4
echo "== dsv3 fp8 w8a8"; timeout 600 python tools/sweep.py --workload dsv3_ep8_rank_fp8w8a8_rows256 --cfgs ";tbmax=1;nt2=2" 2>&1 | grep -v '^{' | tail -3
echo "== int4 m32"; timeout 600 python tools/sweep.py --workload mixtral8x7b_int4g128_decode_m128 --M 32 --cfgs ";nt1=2" 2>&1 | grep -v '^{' | tail -2
echo "== int4 m128"; timeout 600 python tools/sweep.py --workload mixtral8x7b_int4g128_decode_m128 --cfgs ";tiled=-1" 2>&1 | grep -v '^{' | tail -2
