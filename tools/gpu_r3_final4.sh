#!/bin/bash
# Round 3, last evidence session, part 2 (profiles/hbm_traffic.json of part 1 in place): the bench line with the driver's
# flags, the rocprofv3 kernel trace of the bench command, the all-configuration report
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
echo "== bench (driver flags)"; timeout 1200 python bench.py --steps 20 --warmup 5 2>gpurun_out/r03_bench_stderr.log | grep '^{' > gpurun_out/r03_bench_n1.json
python - <<'PY'
import json
j = json.load(open("gpurun_out/r03_bench_n1.json"))
r = j["roofline"]
print("step us", round(j["ms_per_step"] * 1e3, 1), "tok/s", j["value"], r["kernel_ms"], "frac", r["frac"], "stale", r.get("traffic_stale"), "long_run", j.get("long_run"))
c = j.get("cpu_baseline") or {}; print({k: v for k, v in c.items() if k not in ("port", "sample", "host")})
for e in j.get("extra", []):
    r = e["roofline"]
    print(" extra", e["workload"], e["config"]["routing"], "step us", round(e["ms_per_step"]*1e3,1), r["kernel_ms"], "frac", r["frac"], r.get("gemm2"), r.get("traffic"), r.get("traffic_stale"))
PY
echo "== rocprof kernel-trace"; cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_kt -o bench -- python $R/bench.py --no-cpu-baseline --steps 100 > $R/gpurun_out/rocprof_kt.log 2>&1; cd $R
python tools/rocprof_summary.py gpurun_out/prof_kt/bench_results.db > gpurun_out/r03_bench_kernel_trace_stats.json; python -c "
import json; d=json.load(open('gpurun_out/r03_bench_kernel_trace_stats.json'))
for k in d['kernels'][:10]: print(k)"
rm -rf gpurun_out/prof_kt
echo "== report"; timeout 1500 python tools/report.py gpurun_out 2>&1 | grep -v amdgpu.ids | grep "^| [1-5]" | cut -d'|' -f2-7
