#!/bin/bash
# fused output quantisation of the gated fp8 GEMM1: parity, bit-equality with the separate pass, A/B by the "fuseq" knob
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
L=gpurun_out/r3_fq.log; : > $L
echo "== pytest" >> $L
timeout 1500 python -m pytest tests/test_gpu_moe.py tests/test_gpu_fullsize.py tests/test_zz4_gpu_reference_glue.py tests/test_gpu_fused_step.py -m gpu -q -x --timeout 900 -k "a8 or w8a8 or fp8 or prefill or config4 or glm or fused" 2>&1 | tail -15 >> $L
W=glm45air_fp8w8a8_prefill_m8192
for i in 1 2 3; do
for t in "" "fuseq=-1"; do
timeout 300 python bench.py --workload $W --no-extras --no-cpu-baseline --steps 40 --warmup 5 ${t:+--tune $t} 2>/dev/null | grep '^{' | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print('$i tune=[%-9s] step us %7.1f  %s' % ('$t', j['ms_per_step']*1e3, j['roofline']['kernel_ms']))" >> $L
done; done
cat $L
