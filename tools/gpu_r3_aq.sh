#!/bin/bash
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1200 python bench.py --steps 20 --warmup 5 2>gpurun_out/r03_bench_stderr.log | grep '^{' > gpurun_out/r03_bench_n1.json
python - <<'PY'
import json
j = json.load(open("gpurun_out/r03_bench_n1.json"))
r = j["roofline"]
print("step us", round(j["ms_per_step"] * 1e3, 1), "tok/s", j["value"], r["kernel_ms"], "frac", r["frac"], "stale", r.get("traffic_stale"), "long_run", j.get("long_run"))
for e in j.get("extra", []):
    r = e["roofline"]
    print(" extra", e["workload"], e["config"]["routing"], "step us", round(e["ms_per_step"]*1e3,1), r["kernel_ms"], "frac", r["frac"], e.get("long_run", {}).get("ms_per_step"))
PY
