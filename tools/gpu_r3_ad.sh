#!/bin/bash
set -u
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_moe.py -m gpu -q -x --timeout 600 -k "fp16_and_swigluoai" 2>&1 | tail -12
