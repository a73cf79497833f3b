#!/bin/bash
# round 3, call Q: LDS conflict counters of the a8w prefill kernel, ep one-rank script, int4 A/B with alternation
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
echo "== ep one rank"; MASTER_ADDR=127.0.0.1 MASTER_PORT=29517 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 PYTHONPATH=$R timeout 600 python tests/ep_rccl_one_rank.py 2>&1 | grep -v "amdgpu.ids\|socket.cpp" | tail -12
echo "== pmc lds"
cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_LDS SQ_LDS_UNALIGNED_STALL SQ_LDS_MEM_VIOLATIONS -d $R/gpurun_out/pmc_l -o p -- python $R/tools/sweep.py --workload glm45air_fp8w8a8_prefill_m8192 --reps 3 --cfgs ";" > $R/gpurun_out/pmc_l.log 2>&1; cd $R
python tools/rocprof_summary.py gpurun_out/pmc_l/p_results.db --pmc > gpurun_out/pmc_l.json 2>/dev/null
python - <<PY | tee gpurun_out/r3_q_pmc_lds.log
import json
d=json.load(open('gpurun_out/pmc_l.json'))
for k in d['pmc']:
    if 'gemm' in k['kernel']: print(k['kernel'][:60], k['counter'], k['mean'], k['mean_dur_ns'], k['dispatches'])
PY
tail -3 gpurun_out/pmc_l.log
rm -rf gpurun_out/pmc_l
for i in 1 2 3; do
echo "== int4 exact round $i"
timeout 600 python tools/sweep.py --workload mixtral8x7b_int4g128_decode_m128 --cfgs ";pd1=4,pd2=4;pf=4;waves=8,pd1=4,pd2=4" 2>&1 | grep -v '^{\|amdgpu.ids\|^#' | cut -c1-120 | tee -a gpurun_out/r3_q_int4_ab.log
echo "== int4 fast round $i"
timeout 600 python tools/sweep.py --workload mixtral8x7b_int4g128_fast_decode_m128 --cfgs ";pd1=4,pd2=4;pf=4;waves=8" 2>&1 | grep -v '^{\|amdgpu.ids\|^#' | cut -c1-120 | tee -a gpurun_out/r3_q_int4_ab.log
done
