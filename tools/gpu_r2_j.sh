#!/bin/bash
# Round 2, session J: fused decode step, second pass (sort phase on one wavefront; GEMM2 + combine with one workgroup per CU)
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== fused-step tests"; timeout 900 python -m pytest tests/test_gpu_fused_step.py tests/test_gpu_routing.py tests/test_gpu_router.py -q -x --timeout 600 2>&1 | tail -3
run() { timeout 300 python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-extras "$@" 2>/dev/null | grep '^{' | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print(round(j['ms_per_step']*1e3,1), 'us', j['roofline']['kernel_ms'], j['config']['geometry'][95:200])"; }
for i in 1 2 3; do
  echo "== headline default";        run
  echo "== headline fuse=-1";        run --tune fuse=-1
  echo "== headline dbg=128 (1 WG/CU)"; run --tune dbg=128
  echo "== headline fuse2 off only (fuse=3)"; run --tune fuse=3
done
echo "== int4 m128 default"; run --workload mixtral8x7b_int4g128_decode_m128
echo "== int4 m128 fuse=-1"; run --workload mixtral8x7b_int4g128_decode_m128 --tune fuse=-1
echo "== fp8a8 m32 default"; run --workload mixtral8x7b_fp8w8a8_decode_m32
echo "== fp8a8 m32 fuse=-1"; run --workload mixtral8x7b_fp8w8a8_decode_m32 --tune fuse=-1
