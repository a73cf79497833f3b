#!/bin/bash
# full GPU test-suite + the headline bench line (the round-end check the driver runs, plus timing)
set -u
echo "== pytest gpu"; timeout 1500 python -m pytest tests -m gpu -q -x --timeout 600 2>&1 | tail -4
echo "== bench"; timeout 900 python bench.py --no-cpu-baseline 2>/dev/null | grep '^{' | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print(round(j['ms_per_step']*1e3,1), 'us', j['value'], j['roofline']['kernel_ms'], j['roofline']['frac'])"
