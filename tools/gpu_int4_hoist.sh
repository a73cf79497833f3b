#!/bin/bash
# int4 tiled kernels: scale multipliers formed once per 128-k unit (g >= 128).  Parity of everything int4, then
# A/B against the previous build (lvllm_amd/liblkm_prev.so), and the GEMM2 geometry (nt2 / pd2) of the 4-bit formats.
set -u
timeout 400 python -m pytest tests/test_gpu_moe.py -m gpu -x -q -k "int4 or quantised or randomised or prefetch or geometries" 2>&1 | tail -3
W=mixtral8x7b_int4g128_decode_m128
bash tools/gpu_ab.sh "liblkm_prev.so liblkm.so" "$W:128:nt2=2;nt2=2,pd2=2" "$W:32:" "$W:512:" 2>&1 | cut -c1-110
for w in mixtral8x7b_mxfp4_decode_m128 mixtral8x7b_nvfp4_decode_m128; do
  echo "== $w"; timeout 200 python tools/sweep.py --workload $w --reps 20 --cfgs ";;nt2=2;nt2=2,pd2=2;;nt2=2,pd2=2" 2>&1 | grep "^\[" | tail -n +2 | cut -c1-110
done
