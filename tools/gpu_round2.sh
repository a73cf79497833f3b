#!/bin/bash
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== smoke (build then smoke in ONE process: liblkm loaded before torch)"; timeout 600 python __graft_entry__.py smoke 2>&1 | tail -4
echo "== runtime maps"; python - <<'PY'
import torch; torch.cuda.init()
from lvllm_amd import _clib
print(_clib.device_info(), _clib.HIP_RUNTIME_PATH)
print(sorted({l.split()[-1] for l in open('/proc/self/maps') if 'amdhip' in l or 'hsa-runtime' in l or 'liblkm' in l}))
PY
echo "== pytest gpu"; timeout 1500 python -m pytest tests -m gpu -q --timeout 600 2>&1 | tail -15
echo "== sweep bf16 m32"; timeout 600 python tools/sweep.py --g1 "1:2:1,1:2:2,1:1:1" --g2 "1:1,1:2,2:1,2:2" > gpurun_out/sweep_bf16_m32.log 2>&1; grep -v '^{' gpurun_out/sweep_bf16_m32.log | tail -12
echo "== sweep bf16 m128"; timeout 600 python tools/sweep.py --M 128 --g1 "1:4:1,1:2:1,2:2:1,2:1:1" --g2 "1:1,1:2,2:1,2:2,4:1" > gpurun_out/sweep_bf16_m128.log 2>&1; grep -v '^{' gpurun_out/sweep_bf16_m128.log | tail -16
echo "== sweep int4 m128"; timeout 600 python tools/sweep.py --workload mixtral8x7b_int4g128_decode_m128 --g1 "1:4:1,1:2:1,2:2:1,2:1:1,1:1:1" --g2 "1:1,1:2,2:1,2:2,4:1,4:2" > gpurun_out/sweep_int4_m128.log 2>&1; grep -v '^{' gpurun_out/sweep_int4_m128.log | tail -20
echo "== sweep qwen m1"; timeout 600 python tools/sweep.py --workload qwen3_30b_a3b_bf16_decode_m1 --g1 "1:1:1,1:1:2,1:1:4,1:1:8,2:1:8" --g2 "1:1,1:2,1:4,1:8,2:4" > gpurun_out/sweep_qwen_m1.log 2>&1; grep -v '^{' gpurun_out/sweep_qwen_m1.log | tail -16
echo "== bench"; timeout 900 python bench.py 2>&1 | tail -1 | tee gpurun_out/bench_n1.json
echo "== rocprof kernel-trace"; cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_kt -o bench -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 100 > $GRAFT_REPO_ROOT/gpurun_out/rocprof_kt.log 2>&1; cd $GRAFT_REPO_ROOT
find gpurun_out/prof_kt -name "*stats*" | head; for f in $(find gpurun_out/prof_kt -name "*kernel_stats.csv"); do head -12 $f; done
echo "== rocprof pmc FETCH_SIZE"; cd /tmp && timeout 900 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $GRAFT_REPO_ROOT/gpurun_out/prof_pmc_fetch -o bench -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-graph --steps 20 --warmup 5 > $GRAFT_REPO_ROOT/gpurun_out/rocprof_pmc.log 2>&1; cd $GRAFT_REPO_ROOT
ls -R gpurun_out/prof_pmc_fetch | head -20
