#!/usr/bin/env python
"""Condense the three rocprofv3 passes of ONE command (profiles/collect_r04.sh: --kernel-trace --stats, --pmc FETCH_SIZE,
--pmc WRITE_SIZE, each in its own run as MI355X_MICROARCH.md prescribes) into profiles/<round>/<name>_rocprof_summary.json:
per kernel (and grid size) the dispatch count, mean duration, and mean HBM bytes per dispatch — 2 x FETCH_SIZE (the gfx950
correction of the guide: wide coalesced reads are reported at half their size) + WRITE_SIZE, both counters in KB.
usage: python profiles/summarize_cfg.py <dir with prof_<name>_{stats,fetch,write}> <name> <out.json> [command text]"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

src, name, out = sys.argv[1], sys.argv[2], sys.argv[3]
cmd = sys.argv[4] if len(sys.argv) > 4 else ""


def newest(pattern):
    g = sorted(glob.glob(os.path.join(src, pattern), recursive=True), key=os.path.getmtime)
    return g[-1] if g else None


def key(r):
    return (r.get("Kernel_Name", "").split("(")[0][:56], r.get("Grid_Size_X") or r.get("Grid_Size") or "?", r.get("Grid_Size_Y") or "1", r.get("Grid_Size_Z") or "1")


dur = defaultdict(list)
kt = newest(f"prof_{name}_stats/**/*kernel_trace.csv")
if kt:
    for r in csv.DictReader(open(kt)):
        try:
            dur[key(r)].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
        except Exception:
            pass
ctr = {}
for c, d in (("FETCH_SIZE", "fetch"), ("WRITE_SIZE", "write")):
    f = newest(f"prof_{name}_{d}/**/*counter_collection.csv")
    acc = defaultdict(list)
    if f:
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") == c:
                acc[key(r)].append(float(r["Counter_Value"]))
    ctr[c] = acc
rows = []
for k in sorted(set(dur) | set(ctr["FETCH_SIZE"]) | set(ctr["WRITE_SIZE"])):
    v = dur.get(k, [])
    fe, wr = ctr["FETCH_SIZE"].get(k, []), ctr["WRITE_SIZE"].get(k, [])
    row = {"kernel": k[0], "grid": [k[1], k[2], k[3]], "dispatches": len(v) or len(fe) or len(wr)}
    if v:
        row.update(mean_us=sum(v) / len(v) / 1e3, min_us=min(v) / 1e3, max_us=max(v) / 1e3, total_us=sum(v) / 1e3)
    if fe:
        row["fetch_bytes_corrected_x2_mean"] = 2.0 * 1024.0 * sum(fe) / len(fe)
    if wr:
        row["write_bytes_mean"] = 1024.0 * sum(wr) / len(wr)
    if fe and wr:
        row["hbm_bytes_per_dispatch"] = row["fetch_bytes_corrected_x2_mean"] + row["write_bytes_mean"]
    rows.append(row)
rows.sort(key=lambda r: -r.get("total_us", 0.0))
json.dump({"command": cmd, "passes": "rocprofv3 --kernel-trace --stats | --pmc FETCH_SIZE | --pmc WRITE_SIZE (three separate runs), --output-format csv",
           "hbm_bytes": "2 x FETCH_SIZE (KB) + WRITE_SIZE (KB), per dispatch", "kernels": rows}, open(out, "w"), indent=1)
for r in rows[:12]:
    print(f"{r['kernel'][:44]:44s} grid {'x'.join(r['grid']):>14s} n={r['dispatches']:4d} mean {r.get('mean_us', float('nan')):9.1f} us  hbm/dispatch {r.get('hbm_bytes_per_dispatch', float('nan')) / 1e6:9.2f} MB")
