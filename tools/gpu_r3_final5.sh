#!/bin/bash
# Round 3, closing session on the final tree: smoke, full GPU suite, FETCH_SIZE passes (-> gpurun_out/hbm_traffic.json),
# then the bench line with the driver's flags against that fresh traffic file
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== smoke"; timeout 600 python __graft_entry__.py smoke 2>&1 | tail -1 | cut -c1-200
echo "== pytest gpu"; timeout 3000 python -m pytest tests -m gpu -q --timeout 900 2>&1 | tail -3 | tee gpurun_out/r03_pytest_gpu.log
echo "== FETCH_SIZE passes"; timeout 1500 python tools/update_hbm_traffic.py gpurun_out 2>&1 | grep -v amdgpu.ids | cut -c1-120
cp gpurun_out/hbm_traffic.json profiles/hbm_traffic.json
echo "== bench (driver flags)"; timeout 1200 python bench.py --steps 20 --warmup 5 2>gpurun_out/r03_bench_stderr.log | grep '^{' > gpurun_out/r03_bench_n1.json
python - <<'PY'
import json
j = json.load(open("gpurun_out/r03_bench_n1.json"))
r = j["roofline"]
print("step us", round(j["ms_per_step"] * 1e3, 1), "tok/s", j["value"], r["kernel_ms"], "frac", r["frac"], "stale", r.get("traffic_stale"), "long_run", j.get("long_run", {}).get("ms_per_step"))
c = j.get("cpu_baseline") or {}; print({k: v for k, v in c.items() if k not in ("port", "sample", "host")})
for e in j.get("extra", []):
    r = e["roofline"]
    print(" extra", e["workload"], e["config"]["routing"], "step us", round(e["ms_per_step"]*1e3,1), r["kernel_ms"], "frac", r["frac"], r.get("gemm2"), r.get("traffic_stale"))
PY
