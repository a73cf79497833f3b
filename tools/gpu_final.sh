#!/bin/bash
# Round-end evidence: tests, smoke, bench, rocprof kernel-trace + PMC, all-config report.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
echo "== device"; nproc; lscpu | grep -E "Model name|Socket|NUMA node\(s\)" | head -3
echo "== smoke"; timeout 600 python __graft_entry__.py smoke 2>&1 | tail -1
echo "== pytest gpu"; timeout 1500 python -m pytest tests -m gpu -q --timeout 600 2>&1 | tail -3
echo "== bench"; timeout 900 python bench.py 2>/dev/null | grep '^{' | tee gpurun_out/bench_n1.json | cut -c1-400
echo "== host path"; timeout 600 python tools/host_path_timing.py 2>&1 | grep -v amdgpu | tail -1 | tee gpurun_out/host_path.log
echo "== hbm read peak"; python tools/hbm_peak.py 2>&1 | grep -v amdgpu.ids | grep "unroll=8" | tee gpurun_out/hbm_read_peak.log
echo "== rocprof kernel-trace"; cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_kt -o bench -- python $R/bench.py --no-cpu-baseline --no-extras --steps 100 > $R/gpurun_out/rocprof_kt.log 2>&1; cd $R
python tools/rocprof_summary.py gpurun_out/prof_kt/bench_results.db > gpurun_out/kernel_trace_stats.json; python -c "
import json; d=json.load(open('gpurun_out/kernel_trace_stats.json'))
for k in d['kernels'][:8]: print(k)"
echo "== rocprof pmc FETCH_SIZE"; cd /tmp && timeout 900 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/prof_pmc_fetch -o bench -- python $R/bench.py --no-cpu-baseline --no-extras --no-graph --steps 20 --warmup 5 > $R/gpurun_out/rocprof_pmc.log 2>&1; cd $R
python tools/rocprof_summary.py gpurun_out/prof_pmc_fetch/bench_results.db --pmc > gpurun_out/pmc_fetch_size.json; python -c "
import json; d=json.load(open('gpurun_out/pmc_fetch_size.json'))
for k in d['pmc']:
    if 'lkm' in k['kernel']: print(k)"
echo "== rocprof pmc WRITE_SIZE"; cd /tmp && timeout 900 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/gpurun_out/prof_pmc_write -o bench -- python $R/bench.py --no-cpu-baseline --no-extras --no-graph --steps 20 --warmup 5 > $R/gpurun_out/rocprof_pmc_w.log 2>&1; cd $R
python tools/rocprof_summary.py gpurun_out/prof_pmc_write/bench_results.db --pmc > gpurun_out/pmc_write_size.json; python -c "
import json; d=json.load(open('gpurun_out/pmc_write_size.json'))
for k in d['pmc']:
    if 'lkm' in k['kernel']: print(k)"
rm -rf gpurun_out/prof_kt gpurun_out/prof_pmc_fetch gpurun_out/prof_pmc_write
echo "== one-rank EP (captured)"; timeout 300 python bench.py --gpus 1 --steps 100 --warmup 10 --force-ep --no-cpu-baseline --no-extras 2>/dev/null | grep '^{' | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print('force-ep:', j['ms_per_step']*1e3, 'us', j['config']['launch'], j['config']['parallelism'])" | tee gpurun_out/ep_one_rank.log
echo "== report"; timeout 2400 python tools/report.py gpurun_out 2>&1 | grep -v amdgpu.ids | grep "^| [1-5]" | cut -d'|' -f2-6
