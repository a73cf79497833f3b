#!/bin/bash
set -u
export TMPDIR=/tmp
echo "== pytest gpu"; timeout 1500 python -m pytest tests -m gpu -q --timeout 600 2>&1 | tail -12
echo "== report"; timeout 1500 python tools/report.py gpurun_out 2>&1 | grep -v amdgpu.ids | grep "^| [1-5]" | cut -d'|' -f2-8,11
