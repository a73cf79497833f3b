#!/usr/bin/env python3
"""Condenses a rocprofv3 (rocpd sqlite) result into a small text/JSON summary for profiles/.

  python tools/rocprof_summary.py gpurun_out/prof_kt/bench_results.db  [--pmc]
Kernel-trace DB -> per-kernel calls / total / average duration (the `top_kernels` view == --stats).
PMC DB          -> per-kernel mean counter values; FETCH_SIZE is also shown corrected the way
                   MI355X_MICROARCH.md (HBM section) prescribes for gfx950: KB -> bytes, x2 for wide
                   coalesced reads (calibrated in-run on torch's elementwise kernels, see below).
"""
import json
import re
import sqlite3
import sys


def short(name: str) -> str:
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    m = re.match(r"(?:void )?([A-Za-z0-9_:]+(?:<[0-9a-z, ]*>)?)", name)
    s = m.group(1) if m else name
    return s[:90]


def main():
    db = sys.argv[1]
    con = sqlite3.connect(db)
    out = {"db": db}
    rows = con.execute("select name,total_calls,total_duration,average,percentage from top_kernels").fetchall()
    out["kernels"] = [{"kernel": short(n), "calls": c, "total_us": round(t, 1), "avg_us": round(a, 3),
                       "pct": round(p, 2)} for n, c, t, a, p in rows]
    if "--pmc" in sys.argv:
        q = ("select kernel_name, counter_name, count(*), avg(value), avg(duration) from counters_collection "
             "group by kernel_name, counter_name")
        pmc = []
        for n, cn, cnt, v, d in con.execute(q):
            e = {"kernel": short(n), "counter": cn, "dispatches": cnt, "mean": round(v, 3), "mean_dur_ns": round(d, 1)}
            if cn == "FETCH_SIZE":
                e["bytes_corrected_x2"] = round(v * 1024 * 2)
            if cn == "WRITE_SIZE":
                e["bytes_raw"] = round(v * 1024)
            pmc.append(e)
        out["pmc"] = pmc
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
