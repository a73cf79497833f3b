#!/bin/bash
# 16-byte epilogue stores of the fp8 prefill kernel: parity, A/B against the 8-byte build, epilogue ablations
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
L=gpurun_out/r3_wide.log; : > $L
echo "== pytest" >> $L
timeout 1500 python -m pytest tests/test_gpu_moe.py tests/test_gpu_fullsize.py tests/test_zz4_gpu_reference_glue.py -m gpu -q -x --timeout 900 -k "a8 or w8a8 or fp8 or prefill or config4 or glm" 2>&1 | tail -3 >> $L
W=glm45air_fp8w8a8_prefill_m8192
for i in 1 2 3; do
for lib in liblkm.so liblkm_narrow.so; do
[ -f lvllm_amd/$lib ] || continue
LKM_LIB_PATH=$PWD/lvllm_amd/$lib timeout 300 python bench.py --workload $W --no-extras --no-cpu-baseline --steps 40 --warmup 5 2>/dev/null | grep '^{' | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print('$i %-18s step us %7.1f  %s' % ('$lib', j['ms_per_step']*1e3, j['roofline']['kernel_ms']))" >> $L
done; done
for t in dbg=512; do
timeout 300 python bench.py --workload $W --no-extras --no-cpu-baseline --steps 30 --warmup 5 --tune $t 2>/dev/null | grep '^{' | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print('tune=[%-8s] step us %7.1f  %s' % ('$t', j['ms_per_step']*1e3, j['roofline']['kernel_ms']))" >> $L
done
cat $L
