#!/bin/bash
# rocprofv3 PMC passes over one workload's kernels (development): per-kernel mean counter values.
#   bash tools/gpu_pmc.sh <workload> [sweep cfg, e.g. "xcd=1"] [counter set: sq | mem | all | occ (wave residency only)]
# Counter passes run separately (SQ has 8 slots, TCC 4; FETCH_SIZE alone takes 3); --pmc is never
# combined with the sys/hip/hsa trace domains.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
WL=${1:-mixtral8x7b_bf16_decode_m32}; CFG=${2:-}; SET=${3:-all}
run() { # name, counters
  cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $2 -d $R/gpurun_out/pmc_$1 -o p -- python $R/tools/sweep.py --workload $WL --reps 3 --cfgs "$CFG;$CFG" > $R/gpurun_out/pmc_$1.log 2>&1; cd $R
  python tools/rocprof_summary.py gpurun_out/pmc_$1/p_results.db --pmc > gpurun_out/pmc_$1.json 2>/dev/null
  python - <<PY
import json
d=json.load(open('gpurun_out/pmc_$1.json'))
for k in d['pmc']:
    if 'gemm' in k['kernel'] or 'router' in k['kernel']: print(k['kernel'][:60], k['counter'], k['mean'], k['mean_dur_ns'], k['dispatches'])
PY
  rm -rf gpurun_out/pmc_$1
}
if [ "$SET" = occ ]; then
run a "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"
fi
if [ "$SET" = sq ] || [ "$SET" = all ]; then
run a "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"
run b "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC"
run c "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAVES"
fi
if [ "$SET" = mem ] || [ "$SET" = all ]; then
run f "FETCH_SIZE"
run t "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum"
run l "SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_LDS"
fi
