#!/bin/bash
# verification after a planner / kernel change: full GPU suite, then the report rows named in $1 re-measured and
# merged into the committed all-config report (profiles/r01_report_all_configs.*)
set -u
mkdir -p gpurun_out
cp profiles/r01_report_all_configs.md gpurun_out/report.md; cp profiles/r01_report_all_configs.jsonl gpurun_out/report.jsonl
echo "== pytest gpu"; timeout 900 python -m pytest tests -m gpu -q -x --timeout 600 2>&1 | tail -3
echo "== report rows: ${1:-all}"
timeout 900 python tools/report.py gpurun_out ${1:-} 2>&1 | grep -v amdgpu.ids | grep "^| [1-5]" | cut -d'|' -f2-5,7
