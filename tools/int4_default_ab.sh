#!/bin/bash
# configs[2] (Mixtral int4-g128 M=128) through bench.py's captured step, same box, alternating: the tile kernel ("pf" = -1), the
# round-5 default (GEMM1 tile kernel + GEMM2 gemm_w4e.h), gemm_w4e.h for both ("pf" = 6) and gemm_w4x.h ("pf" = 5), uniform and
# Zipf.   bash tools/int4_default_ab.sh   (GPU box)
for rep in 1 2; do for r in uniform zipf; do for t in "pf=-1" "" "pf=6,tiled=64" "pf=5,tiled=64"; do
  out=$(python bench.py --workload mixtral8x7b_int4g128_decode_m128 --no-extras --no-cpu-baseline --full-out "" --steps 200 --routing $r ${t:+--tune $t} 2>/dev/null | tail -1)
  echo "$out" | python -c "
import json,sys
j=json.loads(sys.stdin.read()); r=j['roofline']; km=r['kernel_ms']
print('rep $rep $r ${t:-default}'.ljust(36), 'step %.1f us' % (j['ms_per_step']*1e3), 'gemm1 %.1f gemm2 %.1f' % (km['gemm1']*1e3, km['gemm2']*1e3), r['kernel'][:34])"
done; done; done
