#!/usr/bin/env python
"""profiles/traffic_cfgs.json — `roofline.traffic` of the configurations bench.py appends (C1, C3, C5, taps): HBM bytes per launch from
the PMC passes profiles/collect_r06.sh ran for each (separate --pmc FETCH_SIZE / WRITE_SIZE runs, 2 x FETCH_SIZE + WRITE_SIZE, KB
units — MI355X_MICROARCH.md), condensed by profiles/summarize_cfg.py.   usage: python profiles/make_traffic_cfgs.py <dir> <round>
Only the engine's own kernels (elemhip_*) are summed; what the harness copies or fills is not the path's traffic.
  c3    per launch set of 1024 blocks: the three long-partition kernels (fft, mac, ifft), mean per dispatch
  c1    per synchronous elemhip_process call: all elemhip kernels of the native host's run / its calls (4000 timed + 1 warm-up)
  c5    per 512-frame block of the 128-voice graph: all elemhip kernels of the GPU legs / blocks rendered
  taps  per launch set of 256 blocks: all elemhip kernels / launch sets"""
import json
import os
import re
import sys

src, tag = sys.argv[1], sys.argv[2]
here = os.path.dirname(os.path.abspath(__file__))
out_path = os.path.join(here, "traffic_cfgs.json")
out = json.load(open(out_path)) if os.path.exists(out_path) else {}


def load(name):
    p = os.path.join(src, f"{name}_n1_rocprof_summary.json")
    return json.load(open(p)) if os.path.exists(p) else None


def own_total(j):
    rows = [r for r in j["pmc"] if "elemhip_" in r["kernel"] and "hbm_bytes_per_dispatch" in r]
    return sum(r["dispatches"] * r["hbm_bytes_per_dispatch"] for r in rows), {f'{r["kernel"]} [{r["grid_work_items_total"]} work-items{", " + r["part"] if r.get("part") else ""}]':
                                                                               {"dispatches": r["dispatches"], "hbm_bytes_per_dispatch": r["hbm_bytes_per_dispatch"]} for r in rows}


def last_json(name):
    p = os.path.join(src, f"prof_{name}_plain.log")
    if not os.path.exists(p):
        return {}
    for ln in reversed(open(p, errors="replace").read().splitlines()):
        if ln.startswith("{"):
            try:
                return json.loads(ln)
            except Exception:
                pass
    return {}


j = load("c3")
if j:
    per = {}
    for k in ("elemhip_convolve_long_fft", "elemhip_convolve_long_mac", "elemhip_convolve_long_ifft"):
        rows = sorted([r for r in j["pmc"] if k in r["kernel"]], key=lambda r: -r["dispatches"])
        if rows:
            per[k] = {"hbm_bytes_per_dispatch": rows[0]["hbm_bytes_per_dispatch"], "fetch_x2": rows[0]["fetch_bytes_corrected_x2_mean"], "write": rows[0]["write_bytes_mean"]}
    us = {}
    for k in per:
        rows = sorted([r for r in j["kernel_trace"] if k in r["kernel"]], key=lambda r: -r["dispatches"])
        if rows:
            us[k] = rows[0]["mean_us"]
    out["c3"] = {"hbm_bytes_per_launch": sum(v["hbm_bytes_per_dispatch"] for v in per.values()), "per": "launch set of 1024 blocks x 8 channels", "blocks_per_launch": 1024, "channels": 8,
                 "per_kernel": per, "kernel_mean_us": us, "command_own_step_time": j.get("command_own_step_time"), "round": tag}
j = load("c1")
if j:
    total, rows = own_total(j)
    calls = 4001
    out["c1"] = {"hbm_bytes_per_launch": total / calls, "per": "synchronous elemhip_process call (native host, 4000 timed calls + 1 warm-up)", "kernels": rows, "round": tag}
j = load("c5")
if j:
    # the free-running leg renders 64-block launch sets (its block count depends on the run's speed, and a PMC pass is slow: per-dispatch
    # means are what carries over): one set = the voice level's launch + the mixer level's + the batch epilogue, each the most-dispatched
    # row of its kind
    def top(pred):
        rows = sorted([r for r in j["pmc"] if pred(r) and "hbm_bytes_per_dispatch" in r], key=lambda r: -r["dispatches"])
        return rows[0] if rows else None
    voice = top(lambda r: r["kernel"].startswith("elemhip_spec_island") and r["lds_bytes"] > 100000 and r.get("part") == "sets")
    mixer = top(lambda r: r["kernel"].startswith("elemhip_spec_island") and r["lds_bytes"] < 100000 and r["grid_work_items_total"] >= 65536)
    epi = top(lambda r: r["kernel"].startswith("elemhip_epilogue_batch_kernel") and r["grid_work_items_total"] >= 32768)
    parts = {k: v["hbm_bytes_per_dispatch"] for k, v in (("voice level (128 voice islands)", voice), ("mixer level", mixer), ("batch epilogue", epi)) if v}
    out["c5"] = {"hbm_bytes_per_launch": sum(parts.values()), "per": "64-block launch set of the free-running leg (128 live voices)", "blocks_per_launch": 64,
                 "per_kernel": parts, "round": tag}
j = load("taps")
if j:
    total, rows = own_total(j)
    rec = last_json("taps")
    sets = rec.get("launch_sets")
    if sets:
        out["taps"] = {"hbm_bytes_per_launch": total / sets, "per": f"launch set of {rec.get('blocks_per_step')} blocks", "blocks_per_launch": rec.get("blocks_per_step"), "kernels": rows, "round": tag}
json.dump(out, open(out_path, "w"), indent=1)
print(json.dumps({k: v["hbm_bytes_per_launch"] for k, v in out.items()}))
