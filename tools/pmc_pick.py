#!/usr/bin/env python3
"""prints the FETCH_SIZE rows of a tools/rocprof_summary.py --pmc JSON for kernels whose name contains a substring:
python tools/pmc_pick.py summary.json <label> <substring>"""
import json
import sys
d = json.load(open(sys.argv[1]))
for k in d["pmc"]:
    if sys.argv[3] in k["kernel"] and k["counter"] == "FETCH_SIZE":
        print(sys.argv[2], k["kernel"], "launches", k["dispatches"], "FETCH x2", round(k["bytes_corrected_x2"] / 1e9, 3), "GB", "mean", round(k["mean_dur_ns"] / 1e3, 1), "us")
