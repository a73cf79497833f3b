#!/bin/bash
# round 3, call J: full GPU suite + bench line with the new extras
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
echo "== pytest gpu"; timeout 2400 python -m pytest tests -m gpu -q -x --timeout 900 2>&1 | tail -15 | tee gpurun_out/r3_j_pytest.log
echo "== bench"; timeout 900 python bench.py 2>gpurun_out/r3_j_bench_stderr.log | grep '^{' > gpurun_out/r3_j_bench.json; tail -3 gpurun_out/r3_j_bench_stderr.log
python - <<'PY'
import json
j = json.load(open("gpurun_out/r3_j_bench.json"))
print("step us", round(j["ms_per_step"] * 1e3, 1), "tok/s", j["value"], "frac", j["roofline"]["frac"], "mfma_frac", j["roofline"].get("mfma_frac"), "roof", j["roofline"].get("roof_tflops"), j["roofline"].get("frac_of_roof"))
c = j.get("cpu_baseline") or {}; print({k: v for k, v in c.items() if k not in ("port", "sample", "host")})
for e in j.get("extra", []):
    r = e["roofline"]
    print(" extra", e["workload"], e["config"]["routing"], "step us", round(e["ms_per_step"]*1e3,1), "tok/s", e["value"], r["kernel_ms"], "frac", r["frac"], "mfma_frac", r.get("mfma_frac"), "frac_of_roof", r.get("frac_of_roof"))
PY
