#!/bin/bash
# round 2, GPU session A: full GPU suite (incl. the new full-size parity + EP tests), bench N=1 with extras, the
# one-rank expert-parallel step captured vs eager, the tools written blind in round 1
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
echo "== nproc $(nproc)  gpus: $(python -c 'import torch;print(torch.cuda.device_count())')"
echo "== pytest gpu"; timeout 1500 python -m pytest tests -m gpu -q -x --timeout 600 2>&1 | tail -15 | tee gpurun_out/pytest_gpu.log
echo "== smoke"; timeout 300 python __graft_entry__.py smoke 2>&1 | tail -1 | cut -c1-300
echo "== bench N=1 (headline + extras)"; timeout 900 python bench.py 2>gpurun_out/bench_stderr.log | grep '^{' > gpurun_out/bench_n1.json; tail -5 gpurun_out/bench_stderr.log
python - <<'PY'
import json
j = json.load(open("gpurun_out/bench_n1.json"))
print("step us", round(j["ms_per_step"] * 1e3, 1), "tok/s", j["value"], "kernel_ms", j["roofline"]["kernel_ms"], "frac", j["roofline"]["frac"])
c = j.get("cpu_baseline") or {}; print({k: v for k, v in c.items() if k not in ("port", "sample")})
for e in j.get("extra", []):
    print(" extra", e["workload"], "step us", round(e["ms_per_step"]*1e3,1), "tok/s", e["value"], e["roofline"]["kernel_ms"], "frac", e["roofline"]["frac"], e["config"]["launch"], e["config"]["geometry"].split("|",2)[2][:100])
PY
echo "== one-rank EP: captured vs eager, a2a / ar"
for extra in "" "--no-graph" "--ep-mode ar" "--ep-return f32"; do
  timeout 300 python bench.py --gpus 1 --steps 100 --warmup 10 --force-ep --no-cpu-baseline --no-extras $extra 2>>gpurun_out/ep_stderr.log | grep '^{' | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print('force-ep $extra:', j['ms_per_step']*1e3, 'us', j['config']['launch'], j['config']['parallelism'], j['config'].get('exchange'))" | tee -a gpurun_out/ep_one_rank.log
done
tail -3 gpurun_out/ep_stderr.log
echo "== DSv3 configs[3] on one rank through the EP path (E=256 on one GPU, M=256)"
timeout 600 python bench.py --gpus 1 --steps 50 --warmup 5 --force-ep --no-cpu-baseline --no-extras --workload dsv3_fp8w8a8_ep_decode_b256 2>>gpurun_out/ep_stderr.log | grep '^{' | tee gpurun_out/dsv3_ep1.json | cut -c1-600
echo "== EPLB timings"; timeout 600 python tools/eplb_timing.py 2>&1 | tee gpurun_out/eplb_timing.log | tail -20
echo "== cold-cache variant"
for w in mixtral8x7b_bf16_decode_m32 qwen3_30b_a3b_bf16_decode_m1; do
  timeout 600 python bench.py --workload $w --flush-cache --no-cpu-baseline --no-extras 2>/dev/null | grep '^{' | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print(j['config']['workload'], 'warm', j['ms_per_step'], 'ms  cold', j.get('ms_per_step_cold'), 'ms')" | tee -a gpurun_out/cold_cache.log
done
echo "== probe_hazard (second box)"; /opt/rocm/bin/hipcc -O2 --offload-arch=gfx950 tools/probe_hazard.hip -o /tmp/probe_hazard 2>&1 | tail -2; timeout 300 /tmp/probe_hazard 2>&1 | tee gpurun_out/pk_hazard_probe.log | tail -45
echo "== rocprof kernel-trace"; cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_kt -o bench -- python $R/bench.py --no-cpu-baseline --no-extras --steps 100 > $R/gpurun_out/rocprof_kt.log 2>&1; cd $R
python tools/rocprof_summary.py gpurun_out/prof_kt/bench_results.db > gpurun_out/kernel_trace_stats.json; python -c "
import json; d=json.load(open('gpurun_out/kernel_trace_stats.json'))
for k in d['kernels'][:6]: print(k)"
rm -rf gpurun_out/prof_kt
