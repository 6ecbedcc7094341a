"""The LAST stdout line of `bench.py`: one compact JSON record (< 4 KB) the driver can parse whole.

`bench.py` prints every full record (the headline's and each configuration's) as a JSON line of its own BEFORE this one
(tagged ``"record": <name>``) and appends them to ``gpurun_out/bench_records.jsonl`` when that directory exists; the final
line carries only what the contract names — metric / value / unit / n_gpus / steps / warmup / ms_per_step / dtype / config /
roofline / cpu_baseline / parity — and a `configs` map reduced to six numbers per configuration. The protocol this mirrors is
the reference's own: one result line per run (cli/Benchmark.cpp:105-111).

Round 5's line carried six nested sub-records (20 KB); the driver keeps about 8 KB of stdout, so the head — the headline
itself — was cut off and the round had no driver-verified number. `tests/test_headline.py` pins the size bound.
"""
from __future__ import annotations

import json

MAX_LINE_BYTES = 4096


def _num(x, digits=6):
    """Floats at `digits` significant figures (the full-precision figures are on the full record's line)."""
    if isinstance(x, bool) or x is None or isinstance(x, (int, str)):
        return x
    try:
        return float(f"{float(x):.{digits}g}")
    except (TypeError, ValueError):
        return None


def _short(s, n=110):
    if not isinstance(s, str):
        return s
    return s if len(s) <= n else s[: n - 1].rstrip() + "~"


def _pick(d, keys, strlen=110):
    out = {}
    for k in keys:
        if isinstance(d, dict) and k in d and d[k] is not None:
            v = d[k]
            out[k] = _short(v, strlen) if isinstance(v, str) else _num(v) if not isinstance(v, (dict, list)) else v
    return out


def reduce_config(rec):
    """One configuration's record -> {value, ms_per_step, steps, roofline_frac, cpu_value, parity_ok} (+ a few
    configuration-specific latency figures, all scalars)."""
    if not isinstance(rec, dict):
        return {"error": "no record"}
    if "error" in rec and "value" not in rec:
        return {"error": _short(str(rec.get("error")), 80)}
    roof = rec.get("roofline") or {}
    cpu = rec.get("cpu_baseline") or {}
    par = rec.get("parity") or {}
    out = {
        "value": _num(rec.get("value")), "ms_per_step": _num(rec.get("ms_per_step")), "steps": rec.get("steps"),
        "roofline_frac": _num(roof.get("frac"), 4), "traffic": _num(roof.get("traffic"), 4),
        "cpu_value": _num(cpu.get("value")), "cpu_cores": cpu.get("cores"), "cpu_kind": cpu.get("kind"),
        "parity_ok": par.get("ok"),
    }
    # scalars a reader of the line wants beside the rate (names as in the full record)
    for k in ("us_per_block", "us_per_block_step", "commit_to_first_block_ms_p50", "commit_to_first_block_ms_p99",
              "sync_process_us_per_call", "host_delivered_samples_per_s", "instances_256_samples_per_s"):
        if rec.get(k) is not None:
            out[k] = _num(rec[k], 4)
    upc = rec.get("us_per_call")
    if isinstance(upc, dict):
        out["us_per_call_p50"] = _num(upc.get("p50"), 4)
        if cpu.get("us_per_call_p50") is not None:
            out["cpu_us_per_call_p50"] = _num(cpu["us_per_call_p50"], 4)
    fr = rec.get("free_running")
    if isinstance(fr, dict) and fr.get("ratio") is not None:
        out["mutating_over_static"] = _num(fr["ratio"], 4)
    kern = rec.get("kernels")
    if isinstance(kern, dict) and kern.get("interpreter_block_fraction_counted_leg") is not None:
        out["interp_share_counted"] = _num(kern["interpreter_block_fraction_counted_leg"], 3)
    if roof.get("frac_compulsory") is not None:
        out["roofline_frac_compulsory"] = _num(roof["frac_compulsory"], 4)
    return {k: v for k, v in out.items() if v is not None}


def compact(full: dict) -> dict:
    """The headline record (C2, or `--workload c4`) -> the compact record of the final line."""
    cfg = full.get("config") or {}
    roof = full.get("roofline") or {}
    dom = roof.get("dominant_kernel") or {}
    cpu = full.get("cpu_baseline") or {}
    par = full.get("parity") or {}
    out = _pick(full, ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling"), 100)
    out["vs_baseline"] = full.get("vs_baseline")
    out.update(_pick(full, ("dtype", "data")))
    c = _pick(cfg, ("workload", "blocks_per_step", "mode", "ranks_seen", "voices_per_gpu", "voices_total", "nodes_per_gpu",
                    "instances_per_gpu", "instances_total", "collectives"), 118)
    # the offline rate and the synchronous call side by side, so nobody reads the launch-set figure as a realtime one
    sn = full.get("sync_process_native_host") or {}
    if isinstance(sn, dict) and sn.get("us_p50") is not None:
        c["sync_process_us_per_block_p50"] = _num(sn["us_p50"], 4)
    elif full.get("sync_process_us_per_block") is not None:
        c["sync_process_us_per_block_python"] = _num(full["sync_process_us_per_block"], 4)
    if full.get("us_per_block") is not None:
        c["offline_us_per_block"] = _num(full["us_per_block"], 4)
    out["config"] = c
    r = _pick(roof, ("bound", "achieved", "peak", "unit", "frac"))
    r["traffic"] = _num(roof.get("traffic"), 5)
    if dom:
        r["dominant_kernel"] = _pick(dom, ("kernel", "us_per_launch", "achieved", "frac", "blocks_per_launch"))
    elif roof.get("launch_us_per_step"):
        r["launch_us_per_step"] = [_num(x, 5) for x in roof["launch_us_per_step"][:4]]
    for k in ("algorithmic_bytes_per_step", "algorithmic_bytes_per_launch_set", "kernel_time_fraction_of_step"):
        if roof.get(k) is not None:
            r[k] = _num(roof[k], 5)
    out["roofline"] = r
    if cpu:
        out["cpu_baseline"] = _pick(cpu, ("value", "unit", "cores", "kind", "sample"), 100)
    if full.get("speedup_vs_cpu_baseline") is not None:
        out["speedup_vs_cpu_baseline"] = _num(full["speedup_vs_cpu_baseline"], 4)
    mc = full.get("cpu_baseline_all_cores") or {}
    if isinstance(mc, dict) and mc.get("value") is not None:
        out["cpu_baseline_all_cores"] = _pick(mc, ("value", "cores", "kind"))
    if par:
        err = full.get("parity_max_abs_err")
        if err is None:
            err = par.get("max_abs_err")
        out["parity"] = {"ok": par.get("ok"), "max_abs_err": _num(err, 4), "tolerance": _num(par.get("tolerance"), 3)}
    dr = full.get("device_resident")
    if isinstance(dr, dict) and dr.get("value") is not None:
        out["device_resident"] = _pick(dr, ("value", "ms_per_step"))
    for k in ("host_delivered", "full_chip_256_jobs"):           # (`--workload c4`: the legs behind the device-resident `value`)
        v = full.get(k)
        if isinstance(v, dict) and v.get("value") is not None:
            out[k] = _pick(v, ("value", "ms_per_step", "steps"))
    cfgs = full.get("configs")
    if isinstance(cfgs, dict):
        red = {k: reduce_config(v) for k, v in cfgs.items() if isinstance(v, dict)}
        for k in ("total_wall_s", "all_parity_ok"):
            if k in cfgs:
                red[k] = _num(cfgs[k], 4)
        out["configs"] = red
    out["records"] = "full records: the JSON lines above this one (\"record\": name) and gpurun_out/bench_records.jsonl"
    return out


def headline_line(full: dict) -> str:
    """The compact record serialised; shrinks itself (drops the optional parts first) if it would not fit MAX_LINE_BYTES."""
    out = compact(full)
    line = json.dumps(out, separators=(", ", ": "))
    for drop in ("records", "device_resident", "cpu_baseline_all_cores"):
        if len(line.encode()) < MAX_LINE_BYTES:
            break
        out.pop(drop, None)
        line = json.dumps(out, separators=(",", ":"))
    if len(line.encode()) >= MAX_LINE_BYTES and isinstance(out.get("configs"), dict):
        keep = ("value", "ms_per_step", "steps", "roofline_frac", "cpu_value", "parity_ok")
        out["configs"] = {k: ({kk: v[kk] for kk in keep if kk in v} if isinstance(v, dict) else v) for k, v in out["configs"].items()}
        line = json.dumps(out, separators=(",", ":"))
    assert len(line.encode()) < MAX_LINE_BYTES, len(line)
    return line


def emit(full: dict, stream=None, records_path=None) -> str:
    """Print the full records (one JSON line each, configurations first, then the headline's own full record), then the
    compact line LAST. Returns the compact line."""
    import sys
    stream = stream or sys.stdout
    lines = []
    cfgs = full.get("configs") if isinstance(full.get("configs"), dict) else {}
    for name, rec in cfgs.items():
        if isinstance(rec, dict):
            lines.append(json.dumps(dict({"record": name}, **rec)))
    head = {k: v for k, v in full.items() if k != "configs"}
    lines.append(json.dumps(dict({"record": "headline_full"}, **head)))
    for ln in lines:
        print(ln, file=stream)
    if records_path:
        try:
            with open(records_path, "a") as f:
                for ln in lines:
                    f.write(ln + "\n")
        except OSError:
            pass
    line = headline_line(full)
    print(line, file=stream, flush=True)
    return line
