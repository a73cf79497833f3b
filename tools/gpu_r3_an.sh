#!/bin/bash
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
for r in uniform zipf; do
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/kt_$r -o p -- python $R/bench.py --no-extras --no-cpu-baseline --steps 300 --warmup 20 --routing $r > $R/gpurun_out/kt_$r.log 2>&1; cd $R
python tools/rocprof_summary.py gpurun_out/kt_$r/p_results.db > gpurun_out/r3_an_kt_$r.json
echo "== $r"; grep '^{' gpurun_out/kt_$r.log | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print('step us', round(j['ms_per_step']*1e3,1), j['roofline']['kernel_ms'])"
python - <<PY
import json
d=json.load(open('gpurun_out/r3_an_kt_$r.json'))
for k in d['kernels'][:12]:
    if 'lkm' in k['kernel'] and k['calls'] > 100: print(k['kernel'][:75], k['calls'], k['avg_us'])
PY
rm -rf gpurun_out/kt_$r
done
