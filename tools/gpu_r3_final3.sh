#!/bin/bash
# Round 3, last evidence session, part 1: smoke, the full GPU suite, FETCH_SIZE passes of every bench workload, the A/B of
# the fp8 prefill kernel's epilogue (8-byte stores + separate quantiser pass against 16-byte stores + fused quantisation),
# the instruction-stream probe's added cases
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== smoke"; timeout 600 python __graft_entry__.py smoke 2>&1 | tail -1 | cut -c1-300
echo "== pytest gpu"; timeout 3000 python -m pytest tests -m gpu -q --timeout 900 2>&1 | tail -4 | tee gpurun_out/r03_pytest_gpu.log
echo "== FETCH_SIZE passes"; timeout 1500 python tools/update_hbm_traffic.py gpurun_out 2>&1 | grep -v amdgpu.ids | cut -c1-300
echo "== epilogue A/B"
W=glm45air_fp8w8a8_prefill_m8192
for i in 1 2; do
for v in "liblkm.so:" "liblkm_narrow.so:fuseq=-1"; do
lib=${v%%:*}; t=${v#*:}
LKM_LIB_PATH=$PWD/lvllm_amd/$lib timeout 300 python bench.py --workload $W --no-extras --no-cpu-baseline --steps 40 --warmup 5 ${t:+--tune $t} 2>/dev/null | grep '^{' | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print('$i %-18s tune=[%-9s] step us %7.1f  %s' % ('$lib', '$t', j['ms_per_step']*1e3, j['roofline']['kernel_ms']))"
done; done | tee gpurun_out/r3_epilogue_ab.log
timeout 300 python tools/probe_mfma_valu.py gpurun_out/r3_probe_mfma_valu_3.log --quick > /dev/null 2>&1; grep "32x32x64" gpurun_out/r3_probe_mfma_valu_3.log | grep "1 waves" | cut -c1-150
