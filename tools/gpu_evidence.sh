#!/bin/bash
# Round-end evidence on the GPU box, one parameterised script (replaces the per-session gpu_final*.sh / gpu_r*_final*.sh of
# rounds 1-3, which live in the git history up to commit 69070ea):
#   gpurun -- 'bash tools/gpu_evidence.sh <tag> [steps...]'      steps (default: all, in this order)
#     smoke     __graft_entry__.smoke()
#     tests     pytest -m gpu (log: gpurun_out/<tag>_pytest_gpu.log)
#     traffic   one --pmc FETCH_SIZE pass per bench workload -> gpurun_out/hbm_traffic.json (copy to profiles/)
#     bench     bench.py with the driver's flags -> gpurun_out/<tag>_bench_n1.json
#     trace     rocprofv3 --kernel-trace --stats of the bench command -> gpurun_out/<tag>_bench_kernel_trace_stats.json
#     trace_headline   the same for the headline workload alone (--no-extras): the kernel averages that must agree with roofline.kernel_ms
#     report    every BASELINE configuration, uniform + Zipf -> gpurun_out/<tag>_report_all_configs.{md,jsonl}
#     ep1       one-rank RCCL step, captured
# --pmc passes are separate rocprofv3 runs and never combined with the sys/hip/hsa trace domains.
set -u
TAG=${1:-rXX}; shift || true
STEPS=${*:-"smoke tests traffic bench trace report ep1"}
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for s in $STEPS; do case $s in
smoke) echo "== smoke"; timeout 600 python __graft_entry__.py smoke 2>&1 | tail -1 | cut -c1-200 ;;
tests) echo "== pytest gpu"; timeout 3000 python -m pytest tests/ -x -q -m gpu 2>&1 | tail -3 | tee gpurun_out/${TAG}_pytest_gpu.log ;;
traffic) echo "== FETCH_SIZE passes"; timeout 1500 python tools/update_hbm_traffic.py gpurun_out 2>&1 | grep -v amdgpu.ids | cut -c1-120
         cp gpurun_out/hbm_traffic.json profiles/hbm_traffic.json ;;
bench) echo "== bench (driver flags)"; timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 5 --full-out gpurun_out/${TAG}_bench_full.json 2>gpurun_out/${TAG}_bench_stderr.log > gpurun_out/${TAG}_bench_stdout.log
  tail -1 gpurun_out/${TAG}_bench_stdout.log > gpurun_out/${TAG}_bench_n1.json
  python - <<PY
import json
lines = [json.loads(l) for l in open("gpurun_out/${TAG}_bench_stdout.log") if l.startswith("{")]
j = lines[-1]
r = j["roofline"]
print("last line bytes", len(json.dumps(j)), "step us", round(j["ms_per_step"] * 1e3, 1), "tok/s", j["value"], "regions", j.get("timed_regions_ms"), r["kernel"], r["kernel_ms"], "frac", r["frac"], "traffic", r.get("traffic"), "stale", r.get("traffic_stale"))
c = j.get("cpu_baseline") or {}; print({k: v for k, v in c.items() if k not in ("port", "sample", "host")})
for e in lines[:-1]:
    r = e["roofline"]
    print(" extra", e["extra_workload"], e["routing"], "step us", round(e["ms_per_step"]*1e3,1), r["kernel"], r["kernel_ms"], "frac", r["frac"], "traffic", r.get("traffic"))
PY
  ;;
trace) echo "== rocprof kernel-trace"; cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_kt -o bench -- python $R/bench.py --no-cpu-baseline --steps 100 > $R/gpurun_out/${TAG}_rocprof_kt.log 2>&1; cd $R
  python tools/rocprof_summary.py gpurun_out/prof_kt/bench_results.db > gpurun_out/${TAG}_bench_kernel_trace_stats.json
  python -c "
import json; d=json.load(open('gpurun_out/${TAG}_bench_kernel_trace_stats.json'))
for k in d['kernels'][:10]: print(k)"
  rm -rf gpurun_out/prof_kt ;;
trace_headline) echo "== rocprof kernel-trace, headline workload alone"; cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_kt_h -o bench -- python $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-extras --no-cpu-baseline > $R/gpurun_out/${TAG}_rocprof_kt_headline.log 2>&1; cd $R
  python tools/rocprof_summary.py gpurun_out/prof_kt_h/bench_results.db > gpurun_out/${TAG}_bench_headline_kernel_trace_stats.json
  python -c "
import json; d=json.load(open('gpurun_out/${TAG}_bench_headline_kernel_trace_stats.json'))
for k in d['kernels'][:4]: print(k)"
  rm -rf gpurun_out/prof_kt_h ;;
report) echo "== report"; timeout 2400 python tools/report.py gpurun_out 2>&1 | grep -v amdgpu.ids | grep "^| [1-5]" | cut -d'|' -f2-7
  for x in md jsonl; do [ -f gpurun_out/report.$x ] && cp gpurun_out/report.$x gpurun_out/${TAG}_report_all_configs.$x; done ;;
ep1) echo "== one-rank EP (captured)"; timeout 300 python bench.py --gpus 1 --steps 100 --warmup 10 --force-ep --no-cpu-baseline --no-extras 2>/dev/null | grep '^{' | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print('force-ep:', j['ms_per_step']*1e3, 'us', j['config']['launch'], j['config']['parallelism'])" | tee gpurun_out/${TAG}_ep_one_rank.log ;;
*) echo "unknown step $s" ;;
esac; done
