#!/bin/bash
# Round 2, session M: FETCH_SIZE passes (separate rocprofv3 --pmc runs) for the bench line's extra workloads, so that
# their roofline.traffic is measured like the headline's (profiles/hbm_traffic.json)
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for wl in mixtral8x7b_fp8w8a8_decode_m32 mixtral8x7b_int4g128_decode_m128 mixtral8x7b_int4g128_fast_decode_m128 dsv3_fp8w8a8_ep_decode_b256; do
  cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/pmc_$wl -o p -- python $R/bench.py --workload $wl --no-cpu-baseline --no-extras --no-graph --steps 20 --warmup 5 > $R/gpurun_out/pmc_$wl.log 2>&1; cd $R
  python tools/rocprof_summary.py gpurun_out/pmc_$wl/p_results.db --pmc > gpurun_out/pmc_fetch_$wl.json 2>/dev/null
  python - <<PY
import json
d=json.load(open('gpurun_out/pmc_fetch_$wl.json'))
for k in d['pmc']:
    if 'lkm::gemm' in k['kernel']: print('$wl', k['kernel'][:90], k['dispatches'], k['mean_dur_ns'], k.get('bytes_corrected_x2'))
PY
  rm -rf gpurun_out/pmc_$wl
  grep '^{' gpurun_out/pmc_$wl.log | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print('  algorithmic gemm1 bytes', j['roofline']['algorithmic_bytes'], 'experts hit', j['roofline']['layer']['experts_hit'])"
done
