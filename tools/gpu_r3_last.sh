#!/bin/bash
# last check of round 3: smoke + the whole GPU suite on the final tree
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
echo "== smoke"; timeout 600 python __graft_entry__.py smoke 2>&1 | tail -1 | cut -c1-200
echo "== pytest gpu"; timeout 3000 python -m pytest tests -m gpu -q --timeout 900 2>&1 | tail -4 | tee gpurun_out/r03_pytest_gpu.log
