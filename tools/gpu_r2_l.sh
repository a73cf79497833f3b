#!/bin/bash
# Round 2, session L: weight image layout A/B -- unit-major (default) vs tile-major (LKM_W_TILE_MAJOR=1), same library
set -u
export TMPDIR=/tmp
echo "== stream probe"; timeout 120 tools/_bin/probe_stream 2>&1 | tail -30
echo "== parity (unit-major): moe + fused + fullsize"; timeout 1200 python -m pytest tests/test_gpu_moe.py tests/test_gpu_fused_step.py tests/test_gpu_fullsize.py tests/test_gpu_router.py -q -x --timeout 900 2>&1 | tail -3
run() { timeout 300 python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-extras "$@" 2>/dev/null | grep '^{' | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print(round(j['ms_per_step']*1e3,1), 'us', j['roofline']['kernel_ms'])"; }
for wl in mixtral8x7b_bf16_decode_m32 mixtral8x7b_fp8w8a8_decode_m32 mixtral8x7b_int4g128_decode_m128 mixtral8x7b_mxfp4_decode_m128 mixtral8x7b_mxfp4_decode_m32 dsv3_ep8_rank_fp8w8a8_rows256 qwen3_30b_a3b_bf16_decode_m1 dsv3_fp8w8a8_ep_decode_b256; do
  for rep in 1 2; do
    echo "== $wl unit-major"; run --workload $wl
    echo "== $wl tile-major"; LKM_W_TILE_MAJOR=1 run --workload $wl
  done
done
for wl in glm45air_fp8w8a8_prefill_m8192 glm45air_bf16_prefill_m8192; do
  echo "== $wl unit-major"; run --workload $wl --steps 30 --warmup 5
  echo "== $wl tile-major"; LKM_W_TILE_MAJOR=1 run --workload $wl --steps 30 --warmup 5
done
