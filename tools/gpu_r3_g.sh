#!/bin/bash
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python tests/a8w_debug2.py > gpurun_out/r3_g.log 2>&1
grep -v "amdgpu.ids" gpurun_out/r3_g.log | cut -c1-400
