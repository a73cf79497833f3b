#!/bin/bash
set -u
mkdir -p gpurun_out
echo "== bench"; timeout 900 python bench.py 2>/dev/null | grep '^{' | tee gpurun_out/bench_n1.json | cut -c1-700
echo "== report"; timeout 1700 python tools/report.py gpurun_out 2>&1 | grep -v amdgpu.ids | grep "^| [1-5]" | cut -d'|' -f2-8
