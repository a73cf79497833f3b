#!/bin/bash
# A/B: library built without packed-fp32 VALU (v_pk_fma_f32 / v_pk_mul_f32 stall beside MFMAs) against the default one
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
L=gpurun_out/r3_nopk.log; : > $L
hipcc -O3 -std=c++17 --offload-arch=gfx950 -I lvllm_amd/csrc tools/probe_int4_unit.hip -o /tmp/pi_pk 2>/dev/null
hipcc -O3 -std=c++17 --offload-arch=gfx950 -I lvllm_amd/csrc -Xclang -target-feature -Xclang -packed-fp32-ops tools/probe_int4_unit.hip -o /tmp/pi_nopk 2>/dev/null
echo "== int4 unit probe, packed fp32 (default)" >> $L; timeout 120 /tmp/pi_pk | grep "3 waves" >> $L
echo "== int4 unit probe, no packed fp32" >> $L; timeout 120 /tmp/pi_nopk | grep "3 waves" >> $L
timeout 300 python tools/probe_mfma_valu.py gpurun_out/r3_probe_mfma_valu_2.log --quick > /dev/null 2>&1
for i in 1 2; do
for w in mixtral8x7b_int4g128_decode_m128 mixtral8x7b_int4g128_fast_decode_m128 mixtral8x7b_nvfp4_decode_m128 mixtral8x7b_mxfp4_decode_m128 mixtral8x7b_fp8w8a8_decode_m32 dsv3_fp8w8a8_ep_decode_b256 mixtral8x7b_bf16_decode_m32; do
for lib in liblkm.so liblkm_nopk.so; do
LKM_LIB_PATH=$PWD/lvllm_amd/$lib timeout 300 python bench.py --workload $w --no-extras --no-cpu-baseline --steps 200 --warmup 20 2>/dev/null | grep '^{' | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print('$i %-42s %-16s step us %7.1f  %s' % ('$w', '$lib', j['ms_per_step']*1e3, j['roofline']['kernel_ms']))" >> $L
done; done; done
cat $L
