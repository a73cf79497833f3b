#!/bin/bash
# Round 3 evidence, part 2 (after profiles/hbm_traffic.json of part 1 is in place): the bench line
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== bench"; timeout 1200 python bench.py 2>gpurun_out/r03_bench_stderr.log | grep '^{' > gpurun_out/r03_bench_n1.json
python - <<'PY'
import json
j = json.load(open("gpurun_out/r03_bench_n1.json"))
r = j["roofline"]
print("step us", round(j["ms_per_step"] * 1e3, 1), "tok/s", j["value"], r["kernel_ms"], "frac", r["frac"], "traffic", r.get("traffic"), "stale", r.get("traffic_stale"))
c = j.get("cpu_baseline") or {}; print({k: v for k, v in c.items() if k not in ("port", "sample", "host")})
for e in j.get("extra", []):
    r = e["roofline"]
    print(" extra", e["workload"], e["config"]["routing"], "step us", round(e["ms_per_step"]*1e3,1), r["kernel_ms"], "frac", r["frac"], "traffic", r.get("traffic"), r.get("traffic_stale"))
PY
