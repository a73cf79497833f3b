#!/bin/bash
set -u
export TMPDIR=/tmp
echo "== pytest gpu"; timeout 1500 python -m pytest tests -m gpu -q --timeout 600 2>&1 | tail -5
echo "== report"; timeout 1500 python tools/report.py gpurun_out 2>&1 | grep -v amdgpu.ids | tail -20
