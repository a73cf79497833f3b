#!/bin/bash
# Round 6: configs[2] (Mixtral int4-g128 M=128) through bench.py's captured step, same box, alternating: the round-5 default
# (gemm_w4e.h, "pf" = 6) against the two-queue kernel gemm_w4s.h ("pf" = 7) at weight-ring depths 3 / 4 / 6 / 8 and 4 / 8
# consumer waves, uniform and Zipf.   bash tools/r6_int4_w4s_ab.sh [reps]   (GPU box)
reps=${1:-2}
for rep in $(seq 1 $reps); do for r in uniform zipf; do
  for t in "" "pf=7,tiled=64,pd1=6,pd2=6" "pf=7,tiled=64,pd1=3,pd2=3" "pf=7,tiled=64,pd1=4,pd2=4" "pf=7,tiled=64,pd1=8,pd2=8" \
           "pf=7,tiled=64,pd1=6,pd2=6,waves=8" "pf=7,tiled=64,pd1=4,pd2=4,waves=8" "pf=7,tiled=64,pd1=6,pd2=6,sk2=1" "pf=7,tiled=64,pd1=8,pd2=8,sk2=2"; do
  out=$(python bench.py --workload mixtral8x7b_int4g128_decode_m128 --no-extras --no-cpu-baseline --full-out "" --steps 200 --routing $r ${t:+--tune $t} 2>/dev/null | tail -1)
  echo "$out" | python -c "
import json,sys
try:
    j=json.loads(sys.stdin.read()); r=j['roofline']; km=r['kernel_ms']
    print('rep $rep $r ${t:-default}'.ljust(58), 'step %.1f us' % (j['ms_per_step']*1e3), 'gemm1 %.1f gemm2 %.1f' % (km['gemm1']*1e3, km['gemm2']*1e3), 'frac %.3f' % r['frac'], r['kernel'][:40])
except Exception as e:
    print('rep $rep $r ${t:-default}'.ljust(58), 'FAILED', e)"
done; done; done
