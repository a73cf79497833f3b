#!/usr/bin/env python3
"""Re-measures roofline.traffic for the bench line (on the GPU box): one `rocprofv3 --kernel-trace --pmc FETCH_SIZE` pass
(counters alone, no other trace domain) per workload over `bench.py --workload W --no-graph`, the per-launch mean of the
GEMM1 / GEMM2 kernels corrected as MI355X_MICROARCH.md prescribes for gfx950 (KB -> bytes, x2 for wide coalesced
reads), written to <out>/hbm_traffic.json together with the hash of the kernel sources it was measured on
(bench.py marks the figure stale when the shipped sources differ).
  python tools/update_hbm_traffic.py [out_dir] [workload,workload,...]"""
import json
import os
import shutil
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from bench import EXTRA_EP, EXTRA_N1, HEADLINE, kernel_source_hash  # noqa: E402

OUT = (Path(sys.argv[1]) if len(sys.argv) > 1 else ROOT / "gpurun_out").resolve()


def main():
    # (workload, routing) pairs; an entry of another routing is stored as "workload@routing", the key bench.py looks up
    names = [HEADLINE] + [w if isinstance(w, str) else (f"{w[0]}@{w[1]}" if w[1] != "zipf" else None) for w in EXTRA_N1] + [EXTRA_EP]
    names = [n for n in names if n]
    if len(sys.argv) > 2:
        names = sys.argv[2].split(",")
    OUT.mkdir(parents=True, exist_ok=True)
    res = {"_comment": "HBM bytes per launch from rocprofv3 --pmc FETCH_SIZE (one separate pass per workload: tools/update_hbm_traffic.py), "
                       "KB*1024 and x2 for wide coalesced reads as MI355X_MICROARCH.md prescribes for gfx950.  Each entry names the "
                       "kernel it measured (the name the ENGINE reports for its launches, lkm_last_kernels, matched against the "
                       "profiler's kernel names) and the launch plan; bench.py copies an entry into roofline.traffic only when "
                       "its own kernel and plan are the same, and compares kernel_source_hash with the sources it runs.",
           "kernel_source_hash": kernel_source_hash()}
    prev = OUT / "hbm_traffic.json"
    if len(sys.argv) > 2 and prev.exists():          # partial update: keep the other workloads' entries
        try:
            old = json.loads(prev.read_text())
            if old.get("kernel_source_hash") == res["kernel_source_hash"]:
                res = {**old, **res}
        except Exception:
            pass
    env = dict(os.environ, TMPDIR="/tmp")
    for key in names:
        wl, _, routing = key.partition("@")
        routing = routing or "uniform"
        d = OUT / f"pmc_{wl}_{routing}"
        shutil.rmtree(d, ignore_errors=True)
        # engine defaults (no autotune: a tuning pass would launch candidate kernels of other plans), eager launches
        cmd = ["rocprofv3", "--kernel-trace", "--pmc", "FETCH_SIZE", "-d", str(d), "-o", "p", "--", sys.executable,
               str(ROOT / "bench.py"), "--workload", wl, "--routing", routing, "--no-cpu-baseline", "--no-extras", "--no-graph",
               "--full-line", "--full-out", "", "--steps", "10", "--warmup", "3"]
        r = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True, timeout=900)
        line = next((l for l in reversed(r.stdout.splitlines()) if l.startswith("{")), None)
        s = subprocess.run([sys.executable, str(ROOT / "tools" / "rocprof_summary.py"), str(d / "p_results.db"), "--pmc"],
                           capture_output=True, text=True)
        shutil.rmtree(d, ignore_errors=True)
        if line is None or s.returncode != 0:
            print(f"{key}: FAILED {r.stderr[-300:]} {s.stderr[-300:]}")
            continue
        rf = json.loads(line)["roofline"]
        ent = {"source": f"rocprofv3 --kernel-trace --pmc FETCH_SIZE -- python bench.py --workload {wl} --no-graph --steps 10 "
                         f"--routing {routing} (N=1, engine defaults)", "plan": rf.get("plan")}
        pmc = [k for k in json.loads(s.stdout)["pmc"] if k["counter"] == "FETCH_SIZE"]
        for g, want in (("gemm1", rf.get("kernel")), ("gemm2", rf.get("gemm2_kernel"))):
            # the kernel(s) the engine launched for this GEMM (two for a hybrid plan: their bytes add up per step)
            total, found = 0, []
            for name in (want or "").split("+"):
                hit = [k for k in pmc if k["kernel"].replace(" ", "") == name[:90].replace(" ", "")]
                if name and hit:
                    total += sum(h["bytes_corrected_x2"] for h in hit)
                    found.append(name)
            if found and "+".join(found) == want:
                ent[f"{g}_bytes_per_launch"] = total
                ent[f"{g}_kernel"] = want
            else:
                ent[f"{g}_kernel_not_in_trace"] = want
        if "algorithmic_bytes" in rf:
            ent["gemm1_algorithmic_bytes"] = int(rf["algorithmic_bytes"])
        res[key] = ent
        print(key, ent, flush=True)
    (OUT / "hbm_traffic.json").write_text(json.dumps(res, indent=1) + "\n")


if __name__ == "__main__":
    main()
