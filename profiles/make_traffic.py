#!/usr/bin/env python
"""profiles/traffic*.json (what bench.py reports as `roofline.traffic`: HBM bytes of one launch set from the PMC counters) out of
the per-command summaries profiles/summarize_cfg.py writes.   usage: python profiles/make_traffic.py profiles/r05 r05
Per kernel: 2 x FETCH_SIZE + WRITE_SIZE per dispatch (MI355X_MICROARCH.md: separate --pmc passes, KB units, the gfx950 fetch
correction). Only the kernels the engine launches once per launch set are summed; copies made by the harness are listed apart."""
import json
import os
import sys

src, tag = sys.argv[1], sys.argv[2]
here = os.path.dirname(os.path.abspath(__file__))


def load(name):
    p = os.path.join(src, f"{name}_n1_rocprof_summary.json")
    return json.load(open(p)) if os.path.exists(p) else None


def pick(rows, kernel, part=None, lds=None):
    out = [r for r in rows if r["kernel"].startswith(kernel) and (part is None or r.get("part") == part) and (lds is None or r.get("lds_bytes") == lds)]
    return out


def mean_us(j, kernel, part=None):
    rows = pick(j["kernel_trace"], kernel, part)
    rows.sort(key=lambda r: -r["total_us"])
    return rows[0]["mean_us"] if rows else None


c2 = load("c2")
if c2:
    sets = [r for r in c2["pmc"] if r["kernel"].startswith("elemhip_spec_island") and r.get("hbm_bytes_per_dispatch", 0) > 1e8]
    sets.sort(key=lambda r: -r["hbm_bytes_per_dispatch"])
    epi = [r for r in c2["pmc"] if r["kernel"].startswith("elemhip_epilogue_batch_kernel") and r.get("hbm_bytes_per_dispatch", 0) > 1e6]
    per = {"level 0 elemhip_spec_island (256 voice islands)": sets[0]["hbm_bytes_per_dispatch"],
           "level 1 elemhip_spec_island (2 mixers + root gains, stateless shape)": sets[1]["hbm_bytes_per_dispatch"] if len(sets) > 1 else 0.0,
           "elemhip_epilogue_batch_kernel": epi[0]["hbm_bytes_per_dispatch"] if epi else 0.0}
    total = sum(per.values())
    json.dump({"c2_hbm_bytes_per_block": total / 1024.0, "c2_hbm_bytes_per_launch_set": total, "blocks_per_launch": 1024, "per_kernel": per,
               "fetch_bytes_corrected_x2": sum(r.get("fetch_bytes_corrected_x2_mean", 0.0) for r in sets[:2] + epi[:1]),
               "write_bytes": sum(r.get("write_bytes_mean", 0.0) for r in sets[:2] + epi[:1]),
               "kernel_mean_us": {"level 0": mean_us(c2, "elemhip_spec_island", "sets")},
               "command_own_step_time": c2.get("command_own_step_time"), "round": tag,
               "source": f"profiles/{tag}/c2_n1_rocprof_summary.json: 2 x FETCH_SIZE (gfx950 correction of MI355X_MICROARCH.md) + WRITE_SIZE of the three kernels dispatched "
                         f"once per launch set, separate --pmc passes of `{c2.get('command', '')}`"},
              open(os.path.join(here, "traffic.json"), "w"), indent=1)
    print("traffic.json", total / 1e6, "MB per set")

# (C3 and the other appended configurations: profiles/make_traffic_cfgs.py -> traffic_cfgs.json)
c4 = load("c4")
if c4:
    isl = [r for r in c4["pmc"] if r["kernel"].startswith("elemhip_spec_island") and r.get("part") == "sets"]
    epi = [r for r in c4["pmc"] if r["kernel"].startswith("elemhip_epilogue_batch_kernel") and r.get("dispatches", 0) >= 5]
    cp = [r for r in c4["pmc"] if r["kernel"].startswith("__amd_rocclr_copyBuffer") and r.get("hbm_bytes_per_dispatch", 0) > 1e8]
    # both shapes are dispatched once per set: `dispatches` counts both, the mean is per dispatch
    island = 2.0 * isl[0]["hbm_bytes_per_dispatch"] if isl else 0.0
    e = epi[0]["hbm_bytes_per_dispatch"] if epi else 0.0
    json.dump({"instances": 128, "blocks_per_launch": 1024, "hbm_bytes_per_launch_set": island + e,
               "per_kernel": {"elemhip_spec_island (two shapes, 64 instances each; both dispatches)": island,
                              "elemhip_epilogue_batch_kernel (128 output channels -> the caller's / staging layout)": e},
               "copies_per_launch_set_not_counted": {r["kernel"][:40]: r.get("hbm_bytes_per_dispatch") for r in cp},
               "algorithmic_bytes_per_launch_set": 8589934592, "command_own_step_time": c4.get("command_own_step_time"),
               "kernel_mean_us_under_the_profiler": {"elemhip_spec_island": mean_us(c4, "elemhip_spec_island", "sets"), "elemhip_epilogue_batch_kernel": mean_us(c4, "elemhip_epilogue_batch_kernel")},
               "round": tag, "source": f"profiles/{tag}/c4_n1_rocprof_summary.json: 2 x FETCH_SIZE + WRITE_SIZE per dispatch, separate --pmc passes of `{c4.get('command', '')}`"},
              open(os.path.join(here, "traffic_c4.json"), "w"), indent=1)
    print("traffic_c4.json", (island + e) / 1e6, "MB per set")
