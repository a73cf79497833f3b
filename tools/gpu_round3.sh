#!/bin/bash
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== pytest gpu"; timeout 1500 python -m pytest tests -m gpu -q --timeout 600 2>&1 | tail -25
CF="tiled=-1;tiled=64,waves=8;tiled=64,waves=4;tiled=64,waves=4,nt2=2;tiled=128,waves=8;tiled=128,waves=8,nt2=2"
echo "== bf16 m128"; timeout 600 python tools/sweep.py --M 128 --cfgs "$CF" 2>&1 | grep -v '^{' | tail -8
echo "== bf16 m64"; timeout 600 python tools/sweep.py --M 64 --cfgs "$CF" 2>&1 | grep -v '^{' | tail -8
echo "== bf16 m32 (sort fast path)"; timeout 600 python tools/sweep.py --M 32 --cfgs ";tiled=64,waves=8" 2>&1 | grep -v '^{' | tail -4
echo "== bf16 m512"; timeout 600 python tools/sweep.py --M 512 --cfgs "tiled=64,waves=8;tiled=128,waves=8;tiled=128,waves=8,nt2=2" 2>&1 | grep -v '^{' | tail -5
echo "== int4 m128"; timeout 600 python tools/sweep.py --workload mixtral8x7b_int4g128_decode_m128 --cfgs "$CF" 2>&1 | grep -v '^{' | tail -8
echo "== int4 m32"; timeout 600 python tools/sweep.py --workload mixtral8x7b_int4g128_decode_m128 --M 32 --cfgs ";tiled=64,waves=8" 2>&1 | grep -v '^{' | tail -4
echo "== qwen m1"; timeout 600 python tools/sweep.py --workload qwen3_30b_a3b_bf16_decode_m1 --cfgs ";" 2>&1 | grep -v '^{' | tail -3
