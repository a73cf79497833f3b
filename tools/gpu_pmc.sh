#!/bin/bash
# SQ issue/stall counters for one workload (development): usage gpu_pmc.sh <workload> [extra sweep args]
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
WL=${1:-mixtral8x7b_int4g128_decode_m128}; shift || true
EXTRA="$*"
run() { # name, counters
  cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $2 -d $R/gpurun_out/pmc_$1 -o p -- python $R/tools/sweep.py --workload $WL --reps 5 --cfgs ";" $EXTRA > $R/gpurun_out/pmc_$1.log 2>&1; cd $R
  python tools/rocprof_summary.py gpurun_out/pmc_$1/p_results.db --pmc > gpurun_out/pmc_$1.json 2>/dev/null
  python - <<PY
import json
d=json.load(open('gpurun_out/pmc_$1.json'))
for k in d['pmc']:
    if 'gemm' in k['kernel']: print(k['kernel'][:60], k['counter'], k['mean'], k['mean_dur_ns'])
PY
  rm -rf gpurun_out/pmc_$1
}
run a "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"
run b "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC"
run c "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAVES"
