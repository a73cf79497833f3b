"""bench.py's multi-rank safety net (CPU): when a captured step or an extra workload never completes, rank 0 prints the
line it already holds and every rank leaves with status 0; with nothing to fall back on the status is 3."""
import json
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]


def _run(code: str):
    return subprocess.run([sys.executable, "-c", code], cwd=ROOT, capture_output=True, text=True, timeout=120)


def test_watchdog_prints_the_fallback_line_and_exits_zero():
    r = _run("import time, bench\n"
             "wd = bench.Watchdog(0)\n"
             "wd.arm(1.0, {'metric': 'moe_layer_decode_tokens_per_s', 'value': 1.0}, 'the captured step')\n"
             "time.sleep(30)\n")
    assert r.returncode == 0, r.stderr
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["value"] == 1.0 and "did not complete" in line["note"]
    assert "did not complete in time" in r.stderr


def test_watchdog_disarmed_stays_quiet_and_other_ranks_leave_quietly():
    r = _run("import time, bench\n"
             "wd = bench.Watchdog(0)\n"
             "wd.arm(1.0, {'value': 1.0}, 'x')\n"
             "wd.disarm()\n"
             "time.sleep(3)\n"
             "print('alive')\n")
    assert r.returncode == 0 and r.stdout.strip().endswith("alive") and "{" not in r.stdout
    r = _run("import time, bench\n"
             "wd = bench.Watchdog(3)\n"
             "wd.arm(1.0, {}, 'x')\n"
             "time.sleep(30)\n")
    assert r.returncode == 0 and "{" not in r.stdout      # a non-zero rank holds a marker only and prints nothing
    r = _run("import time, bench\n"
             "wd = bench.Watchdog(0)\n"
             "wd.arm(1.0, None, 'x')\n"
             "time.sleep(30)\n")
    assert r.returncode == 3
