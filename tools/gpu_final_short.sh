#!/bin/bash
# round-end evidence, short form (GPU budget nearly spent): new tests, smoke, the bench line with both CPU baselines,
# and the rocprofv3 kernel trace of the same command on the same box
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
echo "== new tests"; timeout 300 python -m pytest tests -m gpu -q -x --timeout 200 -k "${1:-reference_cpu_kernel}" 2>&1 | tail -2
echo "== smoke"; timeout 300 python __graft_entry__.py smoke 2>&1 | tail -1 | cut -c1-200
echo "== bench"; timeout 600 python bench.py 2>/dev/null | grep '^{' > gpurun_out/bench_n1.json
python - <<'PY'
import json
j = json.load(open("gpurun_out/bench_n1.json"))
print("step us", round(j["ms_per_step"] * 1e3, 1), "tok/s", j["value"], "kernel_ms", j["roofline"]["kernel_ms"], "frac", j["roofline"]["frac"])
c = j["cpu_baseline"]; print({k: v for k, v in c.items() if k not in ("port", "sample")}); print("port:", {k: v for k, v in (c.get("port") or {}).items() if k != "sample"})
PY
echo "== rocprof kernel-trace"; cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_kt -o bench -- python $R/bench.py --no-cpu-baseline --steps 100 > $R/gpurun_out/rocprof_kt.log 2>&1; cd $R
python tools/rocprof_summary.py gpurun_out/prof_kt/bench_results.db > gpurun_out/kernel_trace_stats.json; python -c "
import json; d=json.load(open('gpurun_out/kernel_trace_stats.json'))
for k in d['kernels'][:4]: print(k)"
rm -rf gpurun_out/prof_kt
