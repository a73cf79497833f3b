#!/bin/bash
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
echo "== pytest"; timeout 1800 python -m pytest tests/test_gpu_routing.py tests/test_gpu_moe.py tests/test_gpu_fullsize.py -m gpu -q -x --timeout 900 2>&1 | tail -3
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/kt_pf -o p -- python $R/tools/sweep.py --workload glm45air_fp8w8a8_prefill_m8192 --reps 10 --cfgs ";" > $R/gpurun_out/kt_pf.log 2>&1; cd $R
python tools/rocprof_summary.py gpurun_out/kt_pf/p_results.db > gpurun_out/r3_ae_kt_prefill.json
python - <<'PY'
import json
d=json.load(open('gpurun_out/r3_ae_kt_prefill.json'))
for k in d['kernels'][:24]:
    if 'quant' in k['kernel'] or 'sort' in k['kernel'] or 'items' in k['kernel']: print(k)
PY
rm -rf gpurun_out/kt_pf
grep "auto" gpurun_out/kt_pf.log | cut -c1-120
