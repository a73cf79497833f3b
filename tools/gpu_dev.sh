#!/bin/bash
# Development measurements on the GPU box, one parameterised script (replaces the ~95 one-off gpu_*.sh session scripts of
# rounds 1-3; those are in the git history up to commit 69070ea, their logs under profiles/):
#   gpurun -- 'bash tools/gpu_dev.sh sweep <workload> <M|0> "<cfg;cfg;...>" [reps]'   per-kernel HIP-event times of plans (tools/sweep.py)
#   gpurun -- 'bash tools/gpu_dev.sh pmc   <workload> "<cfg>" [sq|mem|all]'           rocprofv3 counter passes (tools/gpu_pmc.sh)
#   gpurun -- 'bash tools/gpu_dev.sh ab    "<libA> <libB>" "<workload:M:cfgs>" ...'   A/B of library builds (tools/gpu_ab.sh)
#   gpurun -- 'bash tools/gpu_dev.sh test  "<pytest -k expression>" [files...]'       a slice of the GPU suite
# cfg = tuning keys of lkm_set_tuning ("k=v,k=v").  Output: gpurun_out/dev_<mode>.log
set -u
mkdir -p gpurun_out
export PYTHONPATH=$PWD
MODE=${1:-sweep}; shift || true
case $MODE in
sweep) WL=$1; M=$2; CF=$3; REPS=${4:-30}
  timeout 900 python tools/sweep.py --workload $WL $([ "$M" != 0 ] && echo --M $M) --reps $REPS --cfgs "$CF" 2>&1 | grep "^\[\|^#" | cut -c1-260 | tee gpurun_out/dev_sweep.log ;;
pmc) bash tools/gpu_pmc.sh "$1" "${2:-}" "${3:-sq}" 2>&1 | grep -v "^$" | tee gpurun_out/dev_pmc.log ;;
ab) bash tools/gpu_ab.sh "$@" 2>&1 | tee gpurun_out/dev_ab.log ;;
test) K=$1; shift; timeout 1800 python -m pytest ${*:-tests} -m gpu -x -q -k "$K" 2>&1 | tail -15 | tee gpurun_out/dev_test.log ;;
*) echo "unknown mode $MODE" ;;
esac
