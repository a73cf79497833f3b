#!/bin/bash
# Round 3 evidence, part 2: the capacity test in the full suite's position, FETCH_SIZE passes -> hbm_traffic.json, bench line
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
echo "== pytest (ep + capacity, after the moe tests)"; timeout 2400 python -m pytest tests/test_gpu_moe.py tests/test_zz4_gpu_reference_glue.py tests/test_zz5_gpu_create_near_capacity.py -m gpu -q --timeout 900 2>&1 | tail -3
echo "== FETCH_SIZE passes"; timeout 3000 python tools/update_hbm_traffic.py gpurun_out 2>&1 | grep -v amdgpu.ids | cut -c1-400
cp gpurun_out/hbm_traffic.json profiles/hbm_traffic.json
echo "== bench"; timeout 1200 python bench.py 2>gpurun_out/r03_bench_stderr.log | grep '^{' > gpurun_out/r03_bench_n1.json
python - <<'PY'
import json
j = json.load(open("gpurun_out/r03_bench_n1.json"))
r = j["roofline"]
print("step us", round(j["ms_per_step"] * 1e3, 1), "tok/s", j["value"], r["kernel_ms"], "frac", r["frac"], "traffic", r.get("traffic"), "stale", r.get("traffic_stale"))
for e in j.get("extra", []):
    r = e["roofline"]
    print(" extra", e["workload"], e["config"]["routing"], "step us", round(e["ms_per_step"]*1e3,1), r["kernel_ms"], "frac", r["frac"], "traffic", r.get("traffic"), r.get("traffic_stale"))
PY
