#!/bin/bash
# Round 2, session O: three register stages in the fp8 x fp8 streamer (bytes in flight), GEMM2 geometry re-check
set -u
export TMPDIR=/tmp
echo "== parity"; timeout 900 python -m pytest tests/test_gpu_moe.py tests/test_gpu_fullsize.py tests/test_gpu_fused_step.py -q -x --timeout 600 -k "fp8 or w8a8 or a8 or config3 or golden or unit_major" 2>&1 | tail -3
run() { timeout 300 python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-extras "$@" 2>/dev/null | grep '^{' | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print(round(j['ms_per_step']*1e3,1), 'us', j['roofline']['kernel_ms'], j['roofline']['frac'], j['config']['geometry'][100:150])"; }
for i in 1 2; do
  echo "== fp8a8 m32 default"; run --workload mixtral8x7b_fp8w8a8_decode_m32
  echo "== fp8a8 m32 dbg=256 (per-unit scale loads, two waves per SIMD)"; run --workload mixtral8x7b_fp8w8a8_decode_m32 --tune dbg=256
  echo "== fp8a8 m32 nt2=1,sk2=1"; run --workload mixtral8x7b_fp8w8a8_decode_m32 --tune nt2=1,sk2=1
  echo "== fp8a8 m32 nt2=2,sk2=1"; run --workload mixtral8x7b_fp8w8a8_decode_m32 --tune nt2=2,sk2=1
done
echo "== fp8a8 m16"; timeout 300 python tools/sweep.py --workload mixtral8x7b_fp8w8a8_decode_m32 --M 16 --reps 20 --cfgs ";" 2>&1 | grep "^\[" | head -1 | cut -c1-130
echo "== fp8a8 m8"; timeout 300 python tools/sweep.py --workload mixtral8x7b_fp8w8a8_decode_m32 --M 8 --reps 20 --cfgs ";" 2>&1 | grep "^\[" | head -1 | cut -c1-130
