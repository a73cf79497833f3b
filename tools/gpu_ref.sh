#!/bin/bash
# oracle/_ref on the GPU box: smoke (incl. the reference CPU kernel), the GPU suite with durations, the bench line
# with cpu_baseline.kind = "reference".  `bash tools/gpu_ref.sh quick` runs only the *_vs_reference_cpu_kernel tests.
set -u
mkdir -p gpurun_out
if [ "${1:-}" = quick ]; then
  timeout 600 python -m pytest tests -m gpu -q --timeout 300 -k "reference_cpu_kernel" 2>&1 | tail -25
  exit 0
fi
echo "== host"; nproc; lscpu | grep -E "Model name|Socket|NUMA node\(s\)" | head -3
echo "== smoke"; timeout 300 python __graft_entry__.py smoke 2>&1 | tail -1
echo "== pytest gpu (durations)"; timeout 900 python -m pytest tests -m gpu -q -x --timeout 600 --durations=8 2>&1 | tail -14
echo "== bench"; timeout 600 python bench.py 2>gpurun_out/bench_stderr.log | grep '^{' | tee gpurun_out/bench_n1.json | cut -c1-300
python - <<'PY'
import json
j = json.load(open("gpurun_out/bench_n1.json"))
print("step us", round(j["ms_per_step"] * 1e3, 1), "tok/s", j["value"], "frac", j["roofline"]["frac"])
c = j["cpu_baseline"]; print({k: v for k, v in c.items() if k != "port"}); print("port:", c.get("port"))
PY
tail -3 gpurun_out/bench_stderr.log
