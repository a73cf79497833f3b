#!/bin/bash
# round 4, session f: where do the 32x32-MFMA 4-bit kernels win?  default plan vs pf=5 / pf=6 across decode batch sizes
mkdir -p gpurun_out
export PYTHONPATH=$PWD
W=mixtral8x7b_int4g128_decode_m128
for M in 16 32 64 128 256 512; do
  echo "== M=$M"
  CF=";tiled=32;tiled=64;pf=5,tiled=32,waves=4,pd1=2,pd2=2,dbg=1;pf=5,tiled=64,waves=4,pd1=2,pd2=2,dbg=1;pf=6,tiled=32,waves=4,pd1=3,pd2=3,dbg=1;pf=6,tiled=32,waves=7,pd1=3,pd2=3,dbg=1;pf=6,tiled=64,waves=4,pd1=3,pd2=3,dbg=1"
  timeout 300 python tools/sweep.py --workload $W --M $M --reps 30 --cfgs "$CF" 2>&1 | grep "^\[" | tail -n +2 | cut -c1-118
done > gpurun_out/r4f_batch_sweep.log 2>&1
cat gpurun_out/r4f_batch_sweep.log
