#!/bin/bash
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 3000 python tools/report.py gpurun_out 2>&1 | grep -v amdgpu.ids | grep "^| [1-5]" | cut -d'|' -f2-7
