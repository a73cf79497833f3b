#!/bin/bash
# Round 2, session N: fp8 x fp8 streamer with the unit scales of a tile in registers (v_readlane) vs per-unit loads (dbg=256)
set -u
export TMPDIR=/tmp
echo "== parity"; timeout 900 python -m pytest tests/test_gpu_moe.py tests/test_gpu_fullsize.py tests/test_gpu_fused_step.py -q -x --timeout 600 -k "fp8 or w8a8 or a8 or config3 or golden" 2>&1 | tail -3
run() { timeout 300 python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-extras "$@" 2>/dev/null | grep '^{' | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print(round(j['ms_per_step']*1e3,1), 'us', j['roofline']['kernel_ms'], j['roofline']['frac'])"; }
for i in 1 2 3; do
  echo "== fp8a8 m32 unit scales in registers"; run --workload mixtral8x7b_fp8w8a8_decode_m32
  echo "== fp8a8 m32 per-unit scale loads (dbg=256)"; run --workload mixtral8x7b_fp8w8a8_decode_m32 --tune dbg=256
done
