#!/bin/bash
# round 2, GPU session B: the full GPU suite again after the fixes, EPLB timings, the one-rank "ar" capture
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== pytest gpu"; timeout 2000 python -m pytest tests -m gpu -q --timeout 900 2>&1 | tail -25 | tee gpurun_out/pytest_gpu.log
echo "== one-rank EP ar"
timeout 300 python bench.py --gpus 1 --steps 100 --warmup 10 --force-ep --no-cpu-baseline --no-extras --ep-mode ar 2>>gpurun_out/ep_stderr.log | grep '^{' | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print('force-ep ar:', j['ms_per_step']*1e3, 'us', j['config']['launch'], j['config']['parallelism'])" | tee -a gpurun_out/ep_one_rank.log
echo "== EPLB timings"; timeout 600 python tools/eplb_timing.py 2>&1 | tee gpurun_out/eplb_timing.log | tail -20
