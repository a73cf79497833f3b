#!/bin/bash
set -u
mkdir -p gpurun_out
C=${1:-";tiled=256,waves=4,nt1=2,nt2=4;tiled=256,waves=4,nt1=2,nt2=2;tiled=256,waves=8,nt1=1,nt2=4"}
echo "== test"; timeout 600 python -m pytest tests/test_gpu_moe.py -m gpu -q -x --timeout 600 -k "geometries or multi_tile" 2>&1 | tail -2
echo "== glm bf16 prefill"; timeout 900 python tools/sweep.py --workload glm45air_bf16_prefill_m8192 --reps 5 --cfgs "$C" 2>&1 | grep "^\[" | cut -c1-230
