#!/usr/bin/env python
"""Condense rocprofv3 CSV output (profiles/collect.sh) into the per-round summary files under profiles/.

usage: python profiles/summarize.py <gpurun_out dir> <round tag>
Writes gpurun_out/<tag>_summary/ (copied into profiles/<tag>/ and committed from the authoring container).
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

src, tag = sys.argv[1], sys.argv[2]
dst = os.path.join(src, f"{tag}_summary")
os.makedirs(dst, exist_ok=True)


def first(pattern):
    g = sorted(glob.glob(os.path.join(src, pattern), recursive=True), key=os.path.getmtime)
    return g[-1] if g else None        # the newest: a scratch directory may still hold files of earlier runs


# 1. kernel stats (--kernel-trace --stats)
ks = first("prof_stats/**/*kernel_stats.csv")
if ks:
    rows = list(csv.DictReader(open(ks)))
    with open(os.path.join(dst, "c2_n1_kernel_stats.csv"), "w") as f:
        w = csv.DictWriter(f, fieldnames=rows[0].keys())
        w.writeheader()
        w.writerows(rows)
    print("kernel stats:", [(r.get("Name", "")[:40], r.get("Calls"), r.get("AverageNs")) for r in rows[:6]])
# per-dispatch durations by grid size (a level = one grid size) from the kernel trace
kt = first("prof_stats/**/*kernel_trace.csv")
per_grid = defaultdict(list)
if kt:
    for r in csv.DictReader(open(kt)):
        name = r.get("Kernel_Name", "")
        g = (r.get("Grid_Size_X") or r.get("Grid_Size") or "?", r.get("Grid_Size_Y") or "1")
        try:
            per_grid[(name.split("(")[0][:48], g)].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
        except Exception:
            pass
trace_summary = {}
for k, v in per_grid.items():
    name = f"{k[0]}|grid={k[1][0]}x{k[1][1]}"
    big = [x for x in v if x >= 0.5 * max(v)]
    groups = [("", v)] if len(big) == len(v) else [("|sets", big), ("|single blocks", [x for x in v if x < 0.5 * max(v)])]
    for suffix, vv in groups:       # (same split as the PMC passes below: launch sets vs sets of one block)
        trace_summary[name + suffix] = {"dispatches": len(vv), "mean_us": sum(vv) / len(vv) / 1e3, "min_us": min(vv) / 1e3, "max_us": max(vv) / 1e3}

# 2. PMC passes
pmc = {}
for ctr, d in (("FETCH_SIZE", "prof_fetch"), ("WRITE_SIZE", "prof_write")):
    cc = first(f"{d}/**/*counter_collection.csv")
    if not cc:
        continue
    acc = defaultdict(list)
    for r in csv.DictReader(open(cc)):
        if r.get("Counter_Name") != ctr:
            continue
        name = r.get("Kernel_Name", "").split("(")[0][:48]
        g = (r.get("Grid_Size_X") or r.get("Grid_Size") or "?", r.get("Grid_Size_Y") or "1")
        acc[(name, g)].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        # a kernel that renders launch sets AND single blocks with the same grid (the specialised island kernel: elemhip_process
        # uses it as a set of one) is split by counter value: "sets" = the dispatches within 2x of the largest one
        big = [x for x in v if x >= 0.5 * max(v)]
        if len(big) < len(v) and max(v) > 0:
            pmc[f"{k[0]}|grid={k[1][0]}x{k[1][1]}|sets|{ctr}_KB_mean"] = sum(big) / len(big)
            pmc[f"{k[0]}|grid={k[1][0]}x{k[1][1]}|sets|{ctr}_dispatches"] = len(big)
            rest = [x for x in v if x < 0.5 * max(v)]
            pmc[f"{k[0]}|grid={k[1][0]}x{k[1][1]}|single blocks|{ctr}_KB_mean"] = sum(rest) / len(rest)
            pmc[f"{k[0]}|grid={k[1][0]}x{k[1][1]}|single blocks|{ctr}_dispatches"] = len(rest)
        else:
            pmc[f"{k[0]}|grid={k[1][0]}x{k[1][1]}|{ctr}_KB_mean"] = sum(v) / len(v)
            pmc[f"{k[0]}|grid={k[1][0]}x{k[1][1]}|{ctr}_dispatches"] = len(v)

# 3. C3 kernel stats (multi-block convolve kernels)
c3 = first("prof_c3/**/*kernel_stats.csv")
if c3:
    rows = list(csv.DictReader(open(c3)))
    with open(os.path.join(dst, "c3_n1_kernel_stats.csv"), "w") as f:
        w = csv.DictWriter(f, fieldnames=rows[0].keys())
        w.writeheader()
        w.writerows(rows)
    print("c3 kernel stats:", [(r.get("Name", "")[:40], r.get("Calls"), r.get("AverageNs")) for r in rows[:8]])

cmd_file = os.path.join(src, "prof_cmd.txt")
bench_cmd = open(cmd_file).read().strip() if os.path.exists(cmd_file) else "python bench.py (command line not recorded)"
out = {"command": "rocprofv3 {--kernel-trace --stats | --pmc FETCH_SIZE | --pmc WRITE_SIZE} --output-format csv -- " + bench_cmd,
       "kernel_trace_per_dispatch": trace_summary, "pmc_per_dispatch": pmc}
json.dump(out, open(os.path.join(dst, "c2_n1_rocprof_summary.json"), "w"), indent=1)
print(json.dumps(out, indent=1)[:3000])

# 4. HBM traffic of one launch set (MI355X_MICROARCH.md: FETCH_SIZE / WRITE_SIZE are in KB; gfx950 reports wide coalesced reads at
#    half their size, hence 2 x FETCH_SIZE): the kernels dispatched once per launch set = as often as the batch epilogue
def _grid(k):
    try:
        return int(k.split("grid=")[1].split("x")[0])
    except Exception:
        return 0
ep = [(k, v) for k, v in pmc.items() if k.startswith("elemhip_epilogue_batch_kernel") and k.endswith("_dispatches")]
gmax = max([_grid(k) for k, _ in ep] or [0])
# the batch epilogue has one workgroup per block: the large grids are the launch sets (a ragged last set has a smaller one)
sets = sum(v for k, v in ep if k.endswith("FETCH_SIZE_dispatches") and _grid(k) >= gmax // 2)
if sets:
    def per_set(ctr):      # every kernel group dispatched once per launch set (the epilogue's two grids count together)
        kb = 0.0
        for k, v in pmc.items():
            if not k.endswith(ctr + "_dispatches") or "single blocks" in k:
                continue
            mean = pmc[k[:-len("dispatches")] + "KB_mean"]
            if v == sets:
                kb += mean
            elif k.startswith("elemhip_epilogue_batch_kernel") and _grid(k) >= gmax // 2:
                kb += mean * v / sets
        return kb
    fetch_kb, write_kb = per_set("FETCH_SIZE"), per_set("WRITE_SIZE")
    blocks = 1024
    for tok in bench_cmd.split():
        pass
    if "--batch-blocks" in bench_cmd:
        blocks = int(bench_cmd.split("--batch-blocks")[1].split()[0])
    per_set = 2.0 * fetch_kb * 1024.0 + write_kb * 1024.0
    traffic = {"c2_hbm_bytes_per_block": per_set / blocks, "c2_hbm_bytes_per_launch_set": per_set, "blocks_per_launch": blocks,
               "fetch_bytes_corrected_x2": 2.0 * fetch_kb * 1024.0, "write_bytes": write_kb * 1024.0, "round": tag,
               "launch_sets_profiled": sets,
               "source": f"profiles/{tag}/c2_n1_rocprof_summary.json: 2 x FETCH_SIZE (gfx950 correction of MI355X_MICROARCH.md) + WRITE_SIZE over the kernels "
                         f"dispatched once per launch set, separate --pmc passes of `{bench_cmd}`"}
    json.dump(traffic, open(os.path.join(dst, "traffic.json"), "w"), indent=1)
    print("traffic:", traffic["c2_hbm_bytes_per_block"], "bytes per block")
