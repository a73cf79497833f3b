#!/bin/bash
# Round 2, session K: where the 4-bit tile kernels spend their time -- compile-time ablations of the steady loop
# (LKM_ABL in gemm_tiled.h: 4 no barrier, 8 no token staging, 16 no weight loads, 32 no weight decode), results wrong
set -u
export TMPDIR=/tmp
for wl in mixtral8x7b_int4g128_decode_m128 mixtral8x7b_mxfp4_decode_m128; do
  for lib in liblkm.so _abl/liblkm_abl4.so _abl/liblkm_abl8.so _abl/liblkm_abl12.so _abl/liblkm_abl32.so _abl/liblkm_abl44.so _abl/liblkm_abl60.so; do
    echo "== $wl $lib"
    LKM_LIB_PATH=$PWD/lvllm_amd/$lib timeout 300 python tools/sweep.py --workload $wl --M 128 --reps 20 --cfgs ";pd1=4,pd2=4" 2>&1 | grep "^\[" | cut -c1-110
  done
done
