#!/bin/bash
set -u
export TMPDIR=/tmp
echo "== pytest gpu"; timeout 1500 python -m pytest tests -m gpu -q --timeout 600 2>&1 | tail -8
echo "== glm fp8 w8a8"; timeout 600 python tools/sweep.py --workload glm45air_fp8w8a8_prefill_m8192 --reps 5 --cfgs ";tiled=64,waves=4;tiled=128,waves=8" 2>&1 | grep -v '^{' | tail -3
echo "== glm fp8 w8a16"; timeout 600 python tools/sweep.py --workload glm45air_fp8w8a16_prefill_m8192 --reps 5 --cfgs ";" 2>&1 | grep -v '^{' | tail -1
echo "== dsv3 w8a8 zipf"; timeout 600 python bench.py --workload dsv3_ep8_rank_fp8w8a8_rows256 --routing zipf --no-cpu-baseline --steps 100 2>&1 | grep '^{' | python -c "import json,sys; j=json.loads(sys.stdin.read()); print(j['ms_per_step'], j['roofline']['kernel_ms'], j['config']['geometry'])"
