#!/bin/bash
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
hipcc -O3 -std=c++17 --offload-arch=gfx950 -Wno-unused-function -I lvllm_amd/csrc tools/probe_int4_unit.hip -o /tmp/probe_int4_unit 2>/dev/null
timeout 300 /tmp/probe_int4_unit | tee gpurun_out/r3_au_probe_int4_unit.log
