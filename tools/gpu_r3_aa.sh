#!/bin/bash
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
for r in uniform zipf; do
timeout 600 python tools/sweep.py --workload dsv3_fp8w8a8_ep_decode_b256 --routing $r --cfgs ";tiled=64;;tiled=64" 2>&1 | grep -v '^{\|amdgpu.ids\|^#' | cut -c1-110 | tee -a gpurun_out/r3_aa.log
done
timeout 600 python tools/ep_rank_time.py mixtral 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r3_aa_ep_rank_mixtral.jsonl
timeout 900 python tools/ep_rank_time.py dsv3 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r3_aa_ep_rank_dsv3.jsonl
