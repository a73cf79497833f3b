#!/bin/bash
# kernel trace of the fp8 prefill step (which kernels make up the "sort" phase)
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/kt_pf -o p -- python $R/tools/sweep.py --workload glm45air_fp8w8a8_prefill_m8192 --reps 10 --cfgs ";" > $R/gpurun_out/kt_pf.log 2>&1; cd $R
python tools/rocprof_summary.py gpurun_out/kt_pf/p_results.db > gpurun_out/r3_z_kt_prefill.json
python - <<'PY'
import json
d=json.load(open('gpurun_out/r3_z_kt_prefill.json'))
for k in d['kernels'][:22]: print(k)
PY
rm -rf gpurun_out/kt_pf
