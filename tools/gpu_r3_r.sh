#!/bin/bash
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
echo "== ep one rank"; MASTER_ADDR=127.0.0.1 MASTER_PORT=29517 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 PYTHONPATH=$R timeout 600 python tests/ep_rccl_one_rank.py 2>&1 | grep -v "amdgpu.ids\|socket.cpp" | tail -8
cd /tmp && rocprofv3 --list-avail 2>/dev/null | grep -i "ICACHE\|IFETCH\|BARRIER\|SQ_WAIT\|SQ_INST_CYCLES\|SQ_WAVE_DEP\|DEPENDENCY\|SQ_EXP\|SQ_LEVEL\|SQ_BUSY\|VALU_DEP\|STALL" | cut -c1-160 | sort | uniq | head -80 > $R/gpurun_out/r3_r_counters.txt; cd $R
cat gpurun_out/r3_r_counters.txt
echo "== ablations current kernel"
timeout 900 python tools/sweep.py --workload glm45air_fp8w8a8_prefill_m8192 --reps 5 --cfgs ";dbg=1;dbg=2;dbg=9;dbg=17;dbg=25;dbg=65;dbg=89;dbg=3;dbg=67" 2>&1 | grep -v '^{\|amdgpu.ids\|^#' | cut -c1-110 | tee gpurun_out/r3_r_ablations.log
