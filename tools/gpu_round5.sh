#!/bin/bash
set -u
mkdir -p gpurun_out
echo "== pytest gpu"; timeout 1500 python -m pytest tests -m gpu -q -x --timeout 600 2>&1 | tail -3
echo "== report"; timeout 1500 python tools/report.py gpurun_out 2>&1 | grep -v amdgpu.ids | grep "^| [1-5]" | cut -d'|' -f2-7
