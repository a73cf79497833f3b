#!/bin/bash
set -u
export TMPDIR=/tmp
echo "== pytest gpu"; timeout 1500 python -m pytest tests -m gpu -q --timeout 600 2>&1 | tail -6
echo "== glm bf16"; timeout 600 python tools/sweep.py --workload glm45air_bf16_prefill_m8192 --reps 5 --cfgs ";tiled=128,waves=8;tiled=256,waves=8,nt2=2" 2>&1 | grep -v '^{' | tail -3
echo "== mixtral bf16 M=4096 prefill"; timeout 600 python tools/sweep.py --M 4096 --reps 5 --cfgs ";tiled=128,waves=8;tiled=256,waves=8,nt2=2;tiled=64,waves=4" 2>&1 | grep -v '^{' | tail -4
