#!/bin/bash
# Round 2, session I: fused decode step (router + sort in one launch, GEMM2 + combine in one launch)
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== fused-step tests"; timeout 900 python -m pytest tests/test_gpu_fused_step.py -q -x --timeout 600 2>&1 | tail -5
echo "== routing / moe / layer tests"; timeout 1200 python -m pytest tests/test_gpu_routing.py tests/test_gpu_router.py tests/test_gpu_moe.py tests/test_zz2_gpu_layer.py -q -x --timeout 600 2>&1 | tail -3
for t in "" "fuse=-1"; do
  for i in 1 2; do
    echo "== bench headline tune='$t' run $i"
    timeout 300 python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-extras ${t:+--tune $t} 2>/dev/null | grep '^{' | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print(j['ms_per_step']*1e3, 'us', j['roofline']['kernel_ms'], j['config']['geometry'][:120])"
  done
done
for wl in mixtral8x7b_fp8w8a8_decode_m32 mixtral8x7b_int4g128_decode_m128 qwen3_30b_a3b_bf16_decode_m1 dsv3_fp8w8a8_ep_decode_b256; do
  for t in "" "fuse=-1"; do
    echo "== $wl tune='$t'"
    timeout 300 python bench.py --workload $wl --steps 200 --warmup 20 --no-cpu-baseline --no-extras ${t:+--tune $t} 2>/dev/null | grep '^{' | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print(j['ms_per_step']*1e3, 'us', j['roofline']['kernel_ms'], j['config']['geometry'][:140])"
  done
done
echo "== fp8a8 m32 with g2 nt=1 sk=1 (fusable)"
timeout 300 python bench.py --workload mixtral8x7b_fp8w8a8_decode_m32 --steps 200 --warmup 20 --no-cpu-baseline --no-extras --tune nt2=1,sk2=1 2>/dev/null | grep '^{' | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print(j['ms_per_step']*1e3, 'us', j['roofline']['kernel_ms'], j['config']['geometry'][:140])"
