#!/bin/bash
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== GLM fp8-W8A8 prefill ablations (dbg: 1 = no DMA in the loop, 2 = DMA + barriers only, 4 = no L2 prefetch)"
timeout 600 python tools/sweep.py --workload glm45air_fp8w8a8_prefill_m8192 --reps 10 --cfgs "xcd=-1,dbg=2;xcd=-1,dbg=34;xcd=-1,dbg=66;xcd=-1,dbg=98;xcd=1,dbg=2;xcd=1,dbg=34;xcd=1,dbg=98" 2>&1 | grep -v "^{" | cut -c1-150 | tee gpurun_out/glm_a8_ablation.log
