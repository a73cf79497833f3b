#!/bin/bash
set -u
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_quant.py -m gpu -q --timeout 900 2>&1 | grep -v "^$" | tail -30 | cut -c1-300
