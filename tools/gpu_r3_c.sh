#!/bin/bash
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
hipcc -O2 --offload-arch=gfx950 tools/probe_mfma_issue.hip -o /tmp/probe_mfma_issue 2>&1 | grep -E "error" -A3 | head
timeout 300 /tmp/probe_mfma_issue 2>&1 | tee gpurun_out/r3_probe_mfma_issue.log
