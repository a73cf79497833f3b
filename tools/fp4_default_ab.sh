for rep in 1 2; do for wl in mixtral8x7b_mxfp4_decode_m128 mixtral8x7b_nvfp4_decode_m128; do for r in uniform zipf; do for t in "" "pf=6,tiled=64" "pf=5,tiled=64"; do
  out=$(python bench.py --workload $wl --no-extras --no-cpu-baseline --full-out "" --steps 200 --routing $r ${t:+--tune $t} 2>/dev/null | tail -1)
  echo "$out" | python -c "
import json,sys
j=json.loads(sys.stdin.read()); r=j['roofline']; km=r['kernel_ms']
print('rep $rep $wl $r ${t:-default}'.ljust(66), 'step %.1f us' % (j['ms_per_step']*1e3), 'gemm1 %.1f gemm2 %.1f' % (km['gemm1']*1e3, km['gemm2']*1e3), r['kernel'][:30])"
done; done; done; done
