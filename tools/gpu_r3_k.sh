#!/bin/bash
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_zz4_gpu_reference_glue.py -m gpu -q 2>&1 | tail -40 | tee gpurun_out/r3_k_glue.log
