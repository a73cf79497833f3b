#!/bin/bash
# full GPU test suite, then the SQ counter passes on one workload (args: workload, cfg)
set -u
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
bash tools/gpu_pmc.sh "${1:-mixtral8x7b_int4g128_decode_m128}" "${2:-}" sq 2>&1 | grep -v "^$" | cut -c1-170
