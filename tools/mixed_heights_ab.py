#!/usr/bin/env python3
"""A/B of the mixed tile heights (lkm_set_tuning "mixed") on the decode-sized many-expert workloads, uniform and Zipf
routing, through bench.py's captured step (on the GPU box):  python tools/mixed_heights_ab.py [steps]"""
import json
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
STEPS = sys.argv[1] if len(sys.argv) > 1 else "300"
WLS = ["dsv3_ep8_rank_fp8w8a8_rows256", "dsv3_ep8_rank_fp8w8a16_rows256", "dsv3_ep8_rank_bf16_rows256", "dsv3_fp8w8a8_ep_decode_b256"]
SETTINGS = ["mixed=-1", "mixed=0", "mixed=48", "mixed=96", "mixed=-1,tiled=64", "mixed=64,tiled=64"]


def run(wl, routing, tune):
    cmd = [sys.executable, str(ROOT / "bench.py"), "--workload", wl, "--steps", STEPS, "--warmup", "5", "--no-cpu-baseline",
           "--no-extras", "--full-line", "--full-out", "", "--routing", routing, "--tune", tune]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    line = next((l for l in reversed(r.stdout.splitlines()) if l.startswith("{")), None)
    if line is None:
        return None, r.stderr[-300:]
    return json.loads(line), ""


for wl in WLS:
    for routing in ("uniform", "zipf"):
        for tune in SETTINGS:
            j, err = run(wl, routing, tune)
            if j is None:
                print(f"{wl} {routing} {tune}: FAILED {err}", flush=True)
                continue
            km = j["roofline"]["kernel_ms"]
            print(f"{wl} {routing:8s} {tune:20s} step {j['ms_per_step']*1e3:7.1f} us  gemm1 {km['gemm1']*1e3:6.1f} gemm2 {km['gemm2']*1e3:6.1f} "
                  f"frac {j['roofline']['frac']:.3f} experts_hit {j['roofline']['layer']['experts_hit']} | {j['roofline']['plan'].split('| tiled ')[-1][:70]}", flush=True)
