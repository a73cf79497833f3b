#!/bin/bash
set -u
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_moe.py -m gpu -q --timeout 600 -k "mixed_plan" 2>&1 | tail -25
