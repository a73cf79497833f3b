#!/usr/bin/env python3
"""Re-measures roofline.traffic for the bench line (on the GPU box): one `rocprofv3 --kernel-trace --pmc FETCH_SIZE` pass
(counters alone, no other trace domain) per workload over `bench.py --workload W --no-graph`, the per-launch mean of the
GEMM1 / GEMM2 kernels corrected as MI355X_MICROARCH.md prescribes for gfx950 (KB -> bytes, x2 for wide coalesced
reads), written to <out>/hbm_traffic.json together with the hash of the kernel sources it was measured on
(bench.py marks the figure stale when the shipped sources differ).
  python tools/update_hbm_traffic.py [out_dir] [workload,workload,...]"""
import json
import os
import re
import shutil
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from bench import EXTRA_EP, EXTRA_N1, HEADLINE, kernel_source_hash  # noqa: E402

OUT = (Path(sys.argv[1]) if len(sys.argv) > 1 else ROOT / "gpurun_out").resolve()


def gemm_of(kernel: str):
    """'gemm1' / 'gemm2' / None for a kernel name (template arguments: lkm_kernels.h / the kernel headers)"""
    if "gemm1_act_kernel" in kernel:
        return "gemm1"
    if "gemm2_kernel" in kernel or "gemm2_direct_kernel" in kernel:
        return "gemm2"
    m = re.search(r"gemm_prefill_a8w_kernel<\d+, (true|false), (true|false)", kernel)
    if m:
        return "gemm1" if m.group(2) == "true" else "gemm2"
    # gemm_w4x_kernel<WF, ADT, CB, WAVES, GATED, IS_G1, ...> / gemm_w4e_kernel<WF, ADT, CB, NC, GATED, IS_G1, ...>
    m = re.search(r"gemm_w4[xe]_kernel<(?:[^,]+, ){5}(true|false)", kernel)
    if m:
        return "gemm1" if m.group(1) == "true" else "gemm2"
    m = re.search(r"gemm_tiled_kernel<(?:[^,]+, ){6}(true|false)", kernel)
    if m:
        return "gemm1" if m.group(1) == "true" else "gemm2"
    m = re.search(r"gemm_prefill(?:_a8)?_kernel<[^>]*>", kernel)
    if m:
        return "gemm1" if ", true>" in m.group(0) or "true, true" in m.group(0) else "gemm2"
    return None


def main():
    names = [HEADLINE] + [w if isinstance(w, str) else None for w in EXTRA_N1] + [EXTRA_EP]
    names = [n for n in names if n]
    if len(sys.argv) > 2:
        names = sys.argv[2].split(",")
    OUT.mkdir(parents=True, exist_ok=True)
    res = {"_comment": "HBM bytes per launch from rocprofv3 --pmc FETCH_SIZE (one separate pass per workload: tools/update_hbm_traffic.py), "
                       "KB*1024 and x2 for wide coalesced reads as MI355X_MICROARCH.md prescribes for gfx950.  bench.py copies the "
                       "matching entry into roofline.traffic and compares kernel_source_hash with the sources it runs.",
           "kernel_source_hash": kernel_source_hash()}
    env = dict(os.environ, TMPDIR="/tmp")
    for wl in names:
        d = OUT / f"pmc_{wl}"
        shutil.rmtree(d, ignore_errors=True)
        cmd = ["rocprofv3", "--kernel-trace", "--pmc", "FETCH_SIZE", "-d", str(d), "-o", "p", "--", sys.executable,
               str(ROOT / "bench.py"), "--workload", wl, "--no-cpu-baseline", "--no-extras", "--no-graph", "--steps", "10", "--warmup", "3"]
        r = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True, timeout=900)
        line = next((l for l in r.stdout.splitlines() if l.startswith("{")), None)
        s = subprocess.run([sys.executable, str(ROOT / "tools" / "rocprof_summary.py"), str(d / "p_results.db"), "--pmc"],
                           capture_output=True, text=True)
        shutil.rmtree(d, ignore_errors=True)
        if line is None or s.returncode != 0:
            print(f"{wl}: FAILED {r.stderr[-300:]} {s.stderr[-300:]}")
            continue
        j = json.loads(line)
        ent = {"source": f"rocprofv3 --kernel-trace --pmc FETCH_SIZE -- python bench.py --workload {wl} --no-graph --steps 10 (uniform routing, N=1)"}
        for k in json.loads(s.stdout)["pmc"]:
            g = gemm_of(k["kernel"])
            if g and k["counter"] == "FETCH_SIZE":
                # (a step may run two kernels per GEMM -- hybrid plan: the larger one is the dominant kernel)
                if k["bytes_corrected_x2"] > ent.get(f"{g}_bytes_per_launch", 0):
                    ent[f"{g}_bytes_per_launch"] = k["bytes_corrected_x2"]
                    ent[f"{g}_kernel"] = k["kernel"][:80]
        rf = j["roofline"]
        if "algorithmic_bytes" in rf:
            ent["gemm1_algorithmic_bytes"] = int(rf["algorithmic_bytes"])
        res[wl] = ent
        print(wl, ent, flush=True)
    (OUT / "hbm_traffic.json").write_text(json.dumps(res, indent=1) + "\n")


if __name__ == "__main__":
    main()
