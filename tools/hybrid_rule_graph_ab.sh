#!/bin/bash
# The round-5 plan rules (tiles instead of the hybrid) through bench.py's CAPTURED step, same box: default against "hybrid" = 1
# (the round-4 plan).   bash tools/hybrid_rule_graph_ab.sh   (GPU box)
one() {  # workload M routing tune
  out=$(python bench.py --workload $1 --batch-per-gpu $2 --no-extras --no-cpu-baseline --full-out "" --steps 200 --routing $3 ${4:+--tune $4} 2>/dev/null | tail -1)
  echo "$out" | python -c "
import json,sys
j=json.loads(sys.stdin.read()); r=j['roofline']; km=r['kernel_ms']
print('$1 M=$2 $3 ${4:-default}'.ljust(64), 'step %.1f us' % (j['ms_per_step']*1e3), 'gemm1 %.1f gemm2 %.1f' % (km['gemm1']*1e3, km['gemm2']*1e3), j['config']['plan'].split(' | ')[3][:10])"
}
for spec in "dsv3_ep8_rank_fp8w8a8_rows256 32" "dsv3_ep8_rank_fp8w8a16_rows256 32" "dsv3_ep8_rank_fp8w8a16_rows256 40" "mixtral8x7b_fp8w8a8_decode_m32 40" "dsv3_ep8_rank_bf16_rows256 64" "dsv3_ep8_rank_bf16_rows256 128" "qwen3_30b_a3b_bf16_decode_m1 32" "qwen3_30b_a3b_bf16_decode_m1 128" "mixtral8x7b_bf16_decode_m32 48"; do
  set -- $spec
  for r in uniform zipf; do one $1 $2 $r hybrid=1; one $1 $2 $r ""; done
done
for r in uniform zipf; do one mixtral8x7b_int4g128_decode_m128 128 $r pf=-1; one mixtral8x7b_int4g128_decode_m128 128 $r ""; done
