#!/bin/bash
# First GPU pass: tests, smoke, geometry sweep, bench, rocprof.  Everything lands in gpurun_out/.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== device"; rocminfo 2>/dev/null | grep -m3 -E "gfx|Compute Unit" ; nproc; lscpu | grep -E "Model name|Socket|NUMA node\(s\)" 
echo "== smoke"; timeout 600 python __graft_entry__.py smoke 2>&1 | tail -5
echo "== pytest gpu"; timeout 1500 python -m pytest tests -m gpu -q --timeout 600 2>&1 | tail -60
echo "== sweep"; timeout 600 python tools/sweep.py > gpurun_out/sweep_mixtral.log 2>&1; grep -v '^{' gpurun_out/sweep_mixtral.log | tail -40
echo "== bench"; timeout 900 python bench.py 2>&1 | tail -3 | tee gpurun_out/bench_n1.json
