#!/bin/bash
set -u
export TMPDIR=/tmp
echo "== pytest gpu"; timeout 1500 python -m pytest tests -m gpu -q --timeout 600 2>&1 | tail -5
for mode in a2a ar; do
echo "== bench force-ep $mode (1 rank)"; timeout 600 python bench.py --force-ep --ep-mode $mode --steps 100 --no-cpu-baseline 2>&1 | grep '^{' | cut -c1-330
done
