#!/bin/bash
# Round 3 evidence: smoke, full GPU suite, bench line, rocprofv3 kernel trace of the bench command, FETCH_SIZE passes for
# every bench workload (profiles/hbm_traffic.json), PMC table of the fp8 prefill kernels, one-rank EP step, all-config report.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
echo "== device"; nproc; lscpu | grep -E "Model name|Socket|NUMA node\(s\)" | head -3
echo "== smoke"; timeout 600 python __graft_entry__.py smoke 2>&1 | tail -1 | cut -c1-300
echo "== pytest gpu"; timeout 3000 python -m pytest tests -m gpu -q --timeout 900 2>&1 | tail -4 | tee gpurun_out/r03_pytest_gpu.log
echo "== bench"; timeout 1200 python bench.py 2>gpurun_out/r03_bench_stderr.log | grep '^{' > gpurun_out/r03_bench_n1.json
python - <<'PY'
import json
j = json.load(open("gpurun_out/r03_bench_n1.json"))
r = j["roofline"]
print("step us", round(j["ms_per_step"] * 1e3, 1), "tok/s", j["value"], "frac", r["frac"], "mfma_frac", r.get("mfma_frac"), "frac_of_roof", r.get("frac_of_roof"), "traffic", r.get("traffic"), "stale", r.get("traffic_stale"))
c = j.get("cpu_baseline") or {}; print({k: v for k, v in c.items() if k not in ("port", "sample", "host")})
for e in j.get("extra", []):
    r = e["roofline"]
    print(" extra", e["workload"], e["config"]["routing"], "step us", round(e["ms_per_step"]*1e3,1), r["kernel_ms"], "frac", r["frac"], "mfma_frac", r.get("mfma_frac"), "frac_of_roof", r.get("frac_of_roof"))
PY
echo "== rocprof kernel-trace"; cd /tmp && timeout 1200 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_kt -o bench -- python $R/bench.py --no-cpu-baseline --steps 100 > $R/gpurun_out/rocprof_kt.log 2>&1; cd $R
python tools/rocprof_summary.py gpurun_out/prof_kt/bench_results.db > gpurun_out/r03_bench_kernel_trace_stats.json; python -c "
import json; d=json.load(open('gpurun_out/r03_bench_kernel_trace_stats.json'))
for k in d['kernels'][:14]: print(k)"
rm -rf gpurun_out/prof_kt
echo "== FETCH_SIZE passes"; timeout 3000 python tools/update_hbm_traffic.py gpurun_out 2>&1 | grep -v amdgpu.ids | cut -c1-300
echo "== prefill pmc"; bash tools/gpu_pmc.sh glm45air_fp8w8a8_prefill_m8192 "" all 2>&1 | grep "a8w_kernel" | tee gpurun_out/r03_prefill_pmc.log
echo "== one-rank EP (captured)"; timeout 300 python bench.py --gpus 1 --steps 100 --warmup 10 --force-ep --no-cpu-baseline --no-extras 2>/dev/null | grep '^{' | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print('force-ep:', j['ms_per_step']*1e3, 'us', j['config']['launch'], j['config']['parallelism'], j['config'].get('exchange'))" | tee gpurun_out/r03_ep_one_rank.log
echo "== report"; timeout 3000 python tools/report.py gpurun_out 2>&1 | grep -v amdgpu.ids | grep "^| [1-5]" | cut -d'|' -f2-7
