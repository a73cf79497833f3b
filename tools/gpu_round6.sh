#!/bin/bash
set -u
export TMPDIR=/tmp
echo "== hbm read peak"; python tools/hbm_peak.py 2>&1 | grep -v amdgpu.ids | tail -12 | tee gpurun_out/hbm_read_peak.log
echo "== bench force-ep a2a (1 rank)"; timeout 600 python bench.py --force-ep --ep-mode a2a --steps 50 --no-cpu-baseline 2>&1 | grep -v amdgpu.ids | tail -2 | cut -c1-600
echo "== bench force-ep ar (1 rank)"; timeout 600 python bench.py --force-ep --ep-mode ar --steps 50 --no-cpu-baseline 2>&1 | grep -v amdgpu.ids | tail -2 | cut -c1-600
echo "== torchrun 1 proc"; timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29544 bench.py --gpus 1 --steps 50 --warmup 5 --no-cpu-baseline 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-400
echo "== dsv3 rank slice"; timeout 600 python bench.py --workload dsv3_ep8_rank_bf16_rows256 --no-cpu-baseline --steps 100 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-1200
echo "== glm prefill"; timeout 600 python bench.py --workload glm45air_bf16_prefill_m8192 --no-cpu-baseline --steps 20 --warmup 3 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-1200
