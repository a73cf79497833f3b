#!/bin/bash
# PMC passes of the fp8 prefill workload on the final kernels: instruction counts, WRITE_SIZE, wave-cycle split
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
WL=glm45air_fp8w8a8_prefill_m8192
run() { # name, counters
  cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $2 -d $R/gpurun_out/pmc_$1 -o p -- python $R/tools/sweep.py --workload $WL --reps 3 --cfgs ";" > $R/gpurun_out/pmc_$1.log 2>&1; cd $R
  python tools/rocprof_summary.py gpurun_out/pmc_$1/p_results.db --pmc > gpurun_out/pmc_$1.json 2>/dev/null
  python - <<PY
import json
d=json.load(open('gpurun_out/pmc_$1.json'))
for k in d['pmc']:
    if 'a8w' in k['kernel']: print(k['kernel'][:60], k['counter'], k['mean'], k['mean_dur_ns'], k['dispatches'])
PY
  rm -rf gpurun_out/pmc_$1
}
run c "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAVES"
run w "WRITE_SIZE"
run a "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"
