#!/usr/bin/env python
"""The sub-records `bench.py` appends to its line under ``"configs"``: BASELINE.json's other configurations (C1, C3, C4, C5,
the shape-churn variant of C5) and the feedback-tap loop, each TIMED, ROOFLINED, with the reference engine timed beside it on
the host cores of the same box and CHECKED on the samples it just timed (VERDICT r04 "next" #1: every configuration driver-run,
not builder-run).

    python benchmarks/driver_configs.py c1|c3|c5|c5_churn|taps      -> one JSON line on stdout

`bench.py` runs each name in a process of its own (a hang or a crash of one configuration costs that sub-record, never the
headline line); C4 is `bench.py --workload c4` itself. Every record has

    value / unit / steps / ms_per_step          what was timed (inputs resident in HBM when the timed region starts)
    roofline {bound, achieved, peak, unit, frac, basis}    SURVEY.md 8(d) algorithmic bytes of the configuration over the timed region
    cpu_baseline {value, unit, cores, kind, sample}        the reference engine (oracle/_ref; C3: the restatement, labelled)
    parity {ok, max_abs_err, tolerance, what}              bar: 1e-6 abs (x max|ref| when > 1), on the timed samples themselves

The oracle is only ever the checker / the CPU leg here: nothing between a `t0` and its `dt` touches it.
"""
from __future__ import annotations

import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
BLOCK = 512
HBM_PEAK_GBPS = 8000.0


def _pct(a, q):
    a = sorted(a)
    return a[min(len(a) - 1, int(q * len(a)))] if a else None


def _parity(got, ref, what):
    import numpy as np
    peak = float(np.abs(ref).max()) if ref.size else 0.0
    err = float(np.abs(got.astype(np.float64) - ref.astype(np.float64)).max()) if ref.size else float("nan")
    tol = 1e-6 * max(1.0, peak)
    return {"ok": bool(np.isfinite(err) and err <= tol), "max_abs_err": err, "tolerance": tol, "max_abs_ref": peak, "what": what}


def _roofline(alg_bytes_per_block, us_per_block, basis, **extra):
    ach = alg_bytes_per_block / (us_per_block * 1e-6) / 1e9
    d = {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBPS,
         "algorithmic_bytes_per_block": alg_bytes_per_block, "basis": basis, "traffic": None}
    d.update(extra)
    return d


def _traffic(name, **match):
    """`roofline.traffic` of configuration `name`: HBM bytes per launch (PMC passes of this configuration's GPU leg, 2 x FETCH_SIZE +
    WRITE_SIZE as MI355X_MICROARCH.md prescribes; committed under profiles/, not re-measured in this run) when the committed
    geometry matches, else None."""
    try:
        tj = json.load(open(os.path.join(ROOT, "profiles", "traffic_cfgs.json")))[name]
        if all(tj.get(k) == v for k, v in match.items()):
            return tj.get("hbm_bytes_per_launch"), "profiles/traffic_cfgs.json (rocprofv3 PMC passes of this configuration, committed): " + str(tj.get("round", "")) + "; per " + str(tj.get("per", "launch"))
    except Exception:
        pass
    return None, None


def _ref_engine(sr):
    import oracle
    if oracle.have_ref():
        return oracle.RefRuntime(sr, BLOCK, bench_build=os.path.exists(oracle.REF_BENCH_SO)), "reference"
    return oracle.PortRuntime(sr, BLOCK), "port"


# ------------------------------------------------------------------------------------------------------------------------------
# C1 — BASELINE configs[0]: the cli benchmark's own graph and its own protocol (cli/Benchmark.cpp:70-101: one warm-up block,
# N timed synchronous process() calls of 512 frames from a native host)
# ------------------------------------------------------------------------------------------------------------------------------
def _native_run(roots, sr, blocks, spec, dump, resident=False, env_extra=None):
    """examples/bench_cli (C++ over include/elemhip/Runtime.hpp): `blocks` timed calls; every rendered block lands in `dump`."""
    import subprocess
    import tempfile
    from elementary_amd.reconciler import Renderer, batch_to_json
    exe = os.path.join(ROOT, "examples", "bench_cli")
    if not os.path.exists(exe):
        return {"error": "examples/bench_cli is not built"}
    sent = []
    Renderer(lambda b: sent.append(b) or 0).render(*roots)
    with tempfile.TemporaryDirectory() as d:
        bpath = os.path.join(d, "batch.json")
        open(bpath, "w").write(batch_to_json(sent[0]))
        res = subprocess.run([exe, bpath, str(blocks), str(sr), os.path.join(d, "last.f32"), dump], capture_output=True, text=True, timeout=300,
                             env=dict(os.environ, ELEMHIP_SPECIALIZE=str(spec), ELEMHIP_RESIDENT="1" if resident else "0", **(env_extra or {})))
    if res.returncode != 0:
        return {"error": res.stderr[-300:]}
    line = [l for l in res.stderr.splitlines() if l.startswith("{")]
    return json.loads(line[-1]) if line else {"error": "no timing line"}


def c1(calls: int = 4000):
    import tempfile
    import numpy as np
    import torch
    from elementary_amd import graphs
    from elementary_amd.runtime import Runtime
    sr = graphs.C1_SAMPLE_RATE
    roots = graphs.c1_graph()
    alg = graphs.c1_algorithmic_bytes()
    # the reference engine under the same protocol (1 warm-up block + `calls` timed calls), every block kept
    cpu, kind = _ref_engine(sr)
    assert cpu.render(*roots)["result"] == 0
    ref = np.empty((calls + 1, 2, BLOCK), dtype=np.float32)
    ref[0] = cpu.process(None, 2, BLOCK)
    lat = []
    for k in range(calls):
        t0 = time.perf_counter()
        y = cpu.process(None, 2, BLOCK)
        lat.append(time.perf_counter() - t0)
        ref[k + 1] = y
    cpu_us = 1e6 * sum(lat) / calls
    out = {}
    with tempfile.TemporaryDirectory() as d:
        for spec in (2, 1):
            dump = os.path.join(d, f"all{spec}.f32")
            r = _native_run(roots, sr, calls, spec, dump)
            if "error" not in r and os.path.exists(dump):
                got = np.fromfile(dump, dtype=np.float32).reshape(calls + 1, 2, BLOCK)
                r["parity"] = _parity(got, ref, f"every one of the {calls + 1} blocks the native host rendered (warm-up block included) vs the {kind} engine")
            out[spec] = r
        # option `resident` (opt-in): the same protocol with the calls handed to a kernel that stays on the GPU (resident.hip)
        dump = os.path.join(d, "allres.f32")
        r = _native_run(roots, sr, calls, 1, dump, resident=True)
        if "error" not in r and os.path.exists(dump):
            got = np.fromfile(dump, dtype=np.float32).reshape(calls + 1, 2, BLOCK)
            r["parity"] = _parity(got, ref, f"every one of the {calls + 1} blocks the native host rendered with ELEMHIP_RESIDENT=1 vs the {kind} engine")
        out["resident"] = r
    # the same graph through launch sets (what an offline caller gets), output left in HBM
    rt = Runtime(sr, BLOCK, device=0)
    rt.set_option("specialize", 2)
    assert rt.render(*roots)["result"] == 0
    nset, B = 8, 256
    rt.set_option("batch_blocks", B)
    buf = torch.empty((B, 2, BLOCK), dtype=torch.float32, device="cuda")
    rt.process_blocks(B, 2, out_ptr=buf.data_ptr())
    torch.cuda.synchronize()
    keep = []
    t0 = time.perf_counter()
    for _ in range(nset):
        rt.process_blocks(B, 2, out_ptr=buf.data_ptr())
    torch.cuda.synchronize()
    sets_us = 1e6 * (time.perf_counter() - t0) / (nset * B)
    keep = buf.cpu().numpy()                                    # the last timed set
    chk, _ = _ref_engine(sr)
    assert chk.render(*roots)["result"] == 0
    for _ in range(nset * B):
        chk.process(None, 2, BLOCK)
    ref_sets = np.stack([chk.process(None, 2, BLOCK) for _ in range(B)])
    main = out.get(2, {})
    us = main.get("us_mean")
    rec = {
        "config": "BASELINE configs[0] (C1): cli/Benchmark graph, 2 ch el.lowpass(800, 1, el.mul(0.3, el.cycle(440 + c))), 18 nodes, sr 44100, blockSize 512",
        "protocol": "cli/Benchmark.cpp:70-101 on a native host (examples/bench_cli, C++ over the C-ABI): 1 warm-up block, then one synchronous "
                    "elemhip_process call per 512-frame block through host buffers (PCIe + launches + the wait for the block inside every call: the call returns when "
                    "the block's epilogue kernel has published its word to mapped host memory behind the output block, option `sync_poll`)",
        "value": (BLOCK / (us * 1e-6)) if us else None, "unit": "samples/s", "steps": calls, "ms_per_step": (us * 1e-3) if us else None,
        "us_per_call": {"mean": us, "p50": main.get("us_p50"), "p99": main.get("us_p99")}, "specialize": 2,
        "product_default_specialize_1": {k: out.get(1, {}).get(k) for k in ("us_mean", "us_p50", "us_p99", "parity", "error") if k in out.get(1, {})},
        "resident_opt_in": dict({k: out.get("resident", {}).get(k) for k in ("us_mean", "us_p50", "us_p99", "us_max", "parity", "error") if k in out.get("resident", {})},
                                note="option `resident` = 1 (ELEMHIP_RESIDENT=1), product default otherwise: no launch and no stream synchronise per call; "
                                     "the GPU spins on a word in mapped host memory while the host is away (leaves after `resident_idle_us`, default 2000)"),
        "launch_sets": {"us_per_block": sets_us, "samples_per_s": BLOCK / (sets_us * 1e-6), "blocks_per_set": B, "sets_timed": nset,
                        "parity": _parity(keep, ref_sets, f"the last timed set ({B} blocks) vs the {kind} engine advanced through all {(nset + 1) * B} blocks")},
        "roofline": _roofline(alg, us, "algorithmic bytes per block / mean call time; a lone serial phasor -> sin -> svf chain per channel: latency-bound by construction",
                              **dict(zip(("traffic", "traffic_source"), _traffic("c1")))) if us else None,
        "cpu_baseline": {"value": BLOCK / (cpu_us * 1e-6), "unit": "samples/s", "cores": 1, "kind": kind,
                         "sample": f"{calls} synchronous process() calls of the same graph after 1 warm-up block", "us_per_call_mean": cpu_us,
                         "us_per_call_p50": 1e6 * _pct(lat, 0.5)},
        "parity": main.get("parity") or {"ok": False, "error": main.get("error")},
    }
    if us:
        rec["speedup_vs_cpu_baseline"] = cpu_us / us
    return rec


# ------------------------------------------------------------------------------------------------------------------------------
# C3 — BASELINE configs[2]: 8-channel partitioned-FFT convolution reverb, 96 000-tap IRs
# ------------------------------------------------------------------------------------------------------------------------------
def _c3_oracle_channel(args):
    """One channel of C3 on the restatement (oracle/fftconv_oracle.h behind PortRuntime), `blocks` blocks from time zero; the
    first `head` and last `tail` blocks come back."""
    ch, blocks, head, tail = args
    import numpy as np
    import oracle
    from elementary_amd import el, graphs
    rt = oracle.PortRuntime(graphs.C3_SAMPLE_RATE, BLOCK)
    assert rt.add_shared_resource("ir", graphs.c3_impulse_response(ch))
    assert rt.render(el.convolve({"path": "ir"}, el.in_({"channel": 0})))["result"] == 0
    x = graphs.c3_input(graphs.C3_CHANNELS, 64 * BLOCK)[ch].reshape(64, BLOCK)
    first, last = [], []
    t0 = time.perf_counter()
    for b in range(blocks):
        y = rt.process(x[b % 64][None, :], 1, BLOCK)
        if b < head:
            first.append(y[0].copy())
        if b >= blocks - tail:
            last.append(y[0].copy())
    return time.perf_counter() - t0, np.stack(first) if first else None, np.stack(last) if last else None


def _c3_roofline(alg_ref_model, us, set_blocks, ch, parts):
    """C3's roofline, honestly (a fraction above 1 prices a model the kernels do not follow):
      achieved / frac   = the HBM traffic the three kernels of a launch set really move (PMC counters, committed pass of this geometry)
                          / the set's time / 8 TB/s — how busy the memory system is;
      frac_compulsory   = the bytes the long-partition ALGORITHM cannot avoid (input blocks + output blocks + the IR spectra once per
                          set) / time / 8 TB/s — how far the path is from an implementation that moved nothing else;
      reference_model   = SURVEY 8(d)'s figure for the REFERENCE's two-stage convolver (every partition spectrum read once per
                          channel and block) — an aside: this path does not move those bytes."""
    Q = (parts * 512 + 4095) // 4096
    compulsory = 2 * ch * set_blocks * BLOCK * 4 + ch * Q * 4097 * 8          # per launch set
    set_s = us * 1e-6 * set_blocks
    traffic, src = _traffic("c3", blocks_per_launch=set_blocks, channels=ch)
    basis_bytes = traffic if traffic else compulsory
    ach = basis_bytes / set_s / 1e9
    r = {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBPS, "traffic": traffic, "traffic_source": src,
         "basis": ("PMC HBM bytes of one launch set (the three long-partition kernels) / the set's wall time" if traffic else
                   "no committed PMC pass matches this geometry: compulsory bytes of one launch set / the set's wall time"),
         "compulsory_bytes_per_launch_set": compulsory, "frac_compulsory": compulsory / set_s / 1e9 / HBM_PEAK_GBPS,
         "traffic_over_compulsory": (traffic / compulsory) if traffic else None,
         "flops_per_launch_set": {"partition_mac": 8.0 * ch * Q * 4097 * (set_blocks // 8), "transforms_estimate": 2 * ch * (set_blocks // 8) * 5.0 * 4096 * 12},
         "reference_model": {"bytes_per_block": alg_ref_model, "frac_if_priced_against_it": alg_ref_model / (us * 1e-6) / 1e9 / HBM_PEAK_GBPS,
                             "note": "SURVEY 8(d) C3 bytes of the REFERENCE's two-stage partitioning (16 x 513 + 22 x 4097 / 8 bins of 8 B per channel-block + block I/O); "
                                     "not a roofline of these kernels"}}
    return r


def c3(steps: int = 16, warmup: int = 2, set_blocks: int = 1024, gpu_only: bool = False):
    import multiprocessing as mp
    import numpy as np
    import torch
    from elementary_amd import graphs
    from elementary_amd.runtime import Runtime
    ch = graphs.C3_CHANNELS
    rt = Runtime(graphs.C3_SAMPLE_RATE, BLOCK, device=0)
    for c in range(ch):
        assert rt.add_shared_resource(f"ir{c}", graphs.c3_impulse_response(c))
    assert rt.render(*graphs.c3_graph(ch))["result"] == 0
    rt.set_option("batch_blocks", set_blocks)
    for kv in [t for t in os.environ.get("ELEMHIP_C3_OPTS", "").split(",") if t]:      # (A/B runs: "conv_long_mac_lds=0,conv_direct_io=0")
        k_, v_ = kv.split("=", 1)
        rt.set_option(k_, float(v_))
    x = graphs.c3_input(ch, 64 * BLOCK)
    xin = torch.from_numpy(np.ascontiguousarray(x.reshape(ch, 64, BLOCK).transpose(1, 0, 2))).cuda().repeat(max(warmup, steps) * set_blocks // 64, 1, 1).contiguous()
    outs = torch.zeros(((warmup + steps) * set_blocks, ch, BLOCK), dtype=torch.float32, device="cuda")
    stride = set_blocks * ch * BLOCK * 4

    def run(first, n):
        # ONE engine call for the whole stretch, as an offline caller makes it (and as the headline's timed region is one call): the
        # launch sets queue back to back on the engine's stream, no host synchronise between them
        rt.process_blocks(n * set_blocks, ch, out_ptr=outs.data_ptr() + first * stride, in_ptr=xin.data_ptr(), num_inputs=ch)
    run(0, warmup)
    torch.cuda.synchronize()
    # HIP events around the convolve level of every 4th set of the timed region (an event record costs ~4 us of stream time: around every
    # set they were 8.5 of 80 us — profiles/r06/c3_long_mac_variants.txt; ELEMHIP_C3_EVENTS_EVERY=1 / 0 for the A/B)
    rt.set_option("profile_launches", int(os.environ.get("ELEMHIP_C3_EVENTS_EVERY", "4")))
    t0 = time.perf_counter()
    run(warmup, steps)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    prof = rt.launch_profile()
    rt.set_option("profile_launches", 0)
    st = rt.stats()
    us = 1e6 * dt / (steps * set_blocks)
    total = (warmup + steps) * set_blocks
    if gpu_only:      # (profiling passes: the GPU leg only, same launches)
        return {"config": "C3 (GPU leg only)", "value": BLOCK / (us * 1e-6), "unit": "samples/s", "steps": steps, "ms_per_step": 1e3 * dt / steps, "us_per_block": us,
                "launch_us_per_step": [1e3 * v / max(1, prof["launch_sets"]) for v in prof["level_ms"]], "conv_long_sets": rt.describe_plan().get("conv_long_sets")}
    head, tail = 256, 256
    got_head = outs[:head].cpu().numpy().transpose(1, 0, 2)            # [ch][block][frame]
    got_tail = outs[total - tail:].cpu().numpy().transpose(1, 0, 2)
    ctx = mp.get_context("spawn")
    with ctx.Pool(min(ch, max(1, len(os.sched_getaffinity(0))))) as pool:
        res = pool.map(_c3_oracle_channel, [(c, total, head, tail) for c in range(ch)])
    ref_head = np.stack([r[1] for r in res])
    ref_tail = np.stack([r[2] for r in res])
    cpu_s_per_block = sum(r[0] for r in res) / total                     # all 8 channels of one block on ONE core
    p_head = _parity(got_head, ref_head, f"the first {head} blocks of the stream (8 channels) vs the restatement")
    p_tail = _parity(got_tail, ref_tail, f"the LAST {tail} blocks of the last timed launch set (8 channels) vs the restatement advanced through all {total} blocks from time zero")
    alg = graphs.c3_algorithmic_bytes(ch)
    sets = max(1, prof["launch_sets"])
    plan = rt.describe_plan()
    parts = (graphs.C3_IR_LEN + BLOCK - 1) // BLOCK
    # ---- the same graph through the host boundary (VERDICT r05 missing #4): (a) every block delivered to host arrays
    # (elemhip_process_blocks_host, 8 x 2 KB in + out per block over PCIe), (b) one synchronous elemhip_process call per 512-frame block
    # from a native host — what the reference's convolver answers (wasm/Convolve.h:73-84, wasm/Main.cpp:194-216)
    host = {}
    try:
        hb = 4 * set_blocks
        xh = np.ascontiguousarray(np.tile(x, (1, hb // 64)))
        rt.process_blocks_host(xh[:, :set_blocks * BLOCK], ch, set_blocks * BLOCK)                      # warm the staging buffers
        t1 = time.perf_counter()
        yh = rt.process_blocks_host(xh, ch, hb * BLOCK)
        dth = time.perf_counter() - t1
        host["process_blocks_host"] = {"samples_per_s": hb * BLOCK / dth, "us_per_block": 1e6 * dth / hb, "blocks": hb,
                                       "pcie_bytes_per_block": 2 * ch * BLOCK * 4, "delivered_GBps": 2 * ch * BLOCK * 4 * hb / dth / 1e9,
                                       "finite": bool(np.isfinite(yh).all())}
    except Exception as e:      # noqa: BLE001
        host["process_blocks_host"] = {"error": repr(e)[:200]}
    try:
        import tempfile
        with tempfile.TemporaryDirectory() as d:
            res = []
            for c in range(ch):
                pth = os.path.join(d, f"ir{c}.f32")
                graphs.c3_impulse_response(c).astype(np.float32).tofile(pth)
                res.append(f"ir{c}={pth}")
            r = _native_run(graphs.c3_graph(ch), graphs.C3_SAMPLE_RATE, 2000, 1, os.path.join(d, "all.f32"),
                            env_extra={"ELEMHIP_BENCH_RES": ";".join(res), "ELEMHIP_BENCH_IO": f"{ch},{ch}"})
        host["sync_process_native_host"] = r
    except Exception as e:      # noqa: BLE001
        host["sync_process_native_host"] = {"error": repr(e)[:200]}
    wasm = None
    try:
        wj = json.load(open(os.path.join(ROOT, "profiles", "r06", "c3_wasm_reference_cpu.json")))
        wasm = {"value": BLOCK / (wj["us_per_block_mean"] * 1e-6), "unit": "samples/s", "cores": 1, "kind": "reference-wasm", "us_per_block": wj["us_per_block_mean"],
                "sample": f"{wj['blocks_timed']} blocks of the same 8-channel graph on the reference's own wasm engine under Node (benchmarks/c3_wasm_baseline.js)",
                "measured_on": wj.get("measured_on"), "host_cpu": wj.get("host_cpu"),
                "note": "committed figure (profiles/r06/c3_wasm_reference_cpu.json): the reference's wasm build exists only where /root/reference does — not on the GPU box"}
    except Exception:
        pass
    sn = host.get("sync_process_native_host") or {}
    return {
        "config": "BASELINE configs[2] (C3): 8-channel convolution reverb, root(convolve{ir<ch>}(in{ch})), 96 000-tap IRs (2 s at 48 kHz), blockSize 512",
        "protocol": f"elemhip_process_blocks, ONE call for the {steps} timed steps, one step = one launch set of {set_blocks} blocks of all 8 channels; inputs and outputs resident in HBM "
                    "(8 x 2 KB in + out per block: delivery over PCIe would be the bound at this rate)",
        "value": BLOCK / (us * 1e-6), "unit": "samples/s", "steps": steps, "warmup": warmup, "ms_per_step": 1e3 * dt / steps, "us_per_block": us,
        "blocks_per_step": set_blocks, "batch_launches": st["batch_launches"],
        "launch_us_per_step": [1e3 * v / sets for v in prof["level_ms"]], "epilogue_us_per_step": 1e3 * prof["epilogue_ms"] / sets,
        "launch_sets_sampled_by_hip_events": prof["launch_sets"],
        "convolver": {"long_partition_sets": plan.get("conv_long_sets"), "long_partitions_enabled": plan.get("conv_long"), "long_tap_rows": plan.get("conv_max_long_tap_rows"),
                      "note": "sets of a multiple of 8 blocks: the whole IR in 4096-sample partitions (8192-point overlap-save, conv_long.inc), no 512-sample head — "
                              "every input block of a set is known before the convolve level starts"},
        "roofline": _c3_roofline(alg, us, set_blocks, ch, parts),
        "cpu_baseline": {"value": BLOCK / cpu_s_per_block, "unit": "samples/s", "cores": 1, "kind": "port",
                         "sample": f"all {total} blocks of each of the 8 channels on the restatement (oracle/fftconv_oracle.h: the reference library's two-stage "
                                   "partitioning with a plain radix-2 FFT, not Ooura's), channel after channel as one core would; convolve is not in the natively "
                                   "compiled reference (un-vendored FFTConvolver submodule)", "us_per_block": 1e6 * cpu_s_per_block},
        "speedup_vs_cpu_baseline": 1e6 * cpu_s_per_block / us,
        "cpu_baseline_reference_wasm": wasm,
        "host_boundary": host,
        "sync_process_us_per_call": sn.get("us_p50"), "host_delivered_samples_per_s": (host.get("process_blocks_host") or {}).get("samples_per_s"),
        "partition_mac": "vector FMA in registers (elemhip_convolve_long_mac, 0.8 GFLOP per set); the matrix-core Toeplitz kernel "
                         "(elemhip_convolve_batch_mac_mfma, v_mfma_f32_4x4x1) serves sets that are no multiple of 8 blocks and IRs below 32 partitions",
        "parity": {"ok": bool(p_head["ok"] and p_tail["ok"]), "max_abs_err": max(p_head["max_abs_err"], p_tail["max_abs_err"]),
                   "tolerance": min(p_head["tolerance"], p_tail["tolerance"]), "head": p_head, "tail": p_tail,
                   "checker": "restatement, itself pinned to recordings of the reference's wasm engine (tests/golden/convolve_wasm*.f32)"},
    }


# ------------------------------------------------------------------------------------------------------------------------------
# feedback loops through taps (Feedback.h:90-126): 8 filtered loops under two roots
# ------------------------------------------------------------------------------------------------------------------------------
def _tap_graph():
    from elementary_amd import el

    def loop(k, x):
        fb = el.tapIn({"name": f"rv{k}"})
        body = el.lowpass(900.0 + 170.0 * k, 0.9, el.add(x, el.mul(0.7, el.sdelay({"size": 200 + 13 * k}, fb))))
        return el.tanh(el.tapOut({"name": f"rv{k}"}, body))
    x = el.in_({"channel": 0})
    return [el.add(*[loop(k, x) for k in range(0, 8, 2)]), el.add(*[loop(k, x) for k in range(1, 8, 2)])]


def taps(steps: int = 8, warmup: int = 1, set_blocks: int = 256, gpu_only: bool = False):
    import numpy as np
    import torch
    from elementary_amd.runtime import Runtime
    sr = 48000.0
    rt = Runtime(sr, BLOCK, device=0)
    rt.set_option("batch_blocks", set_blocks)
    rt.set_option("specialize", 2)
    assert rt.render(*_tap_graph())["result"] == 0
    rng = np.random.default_rng(7)
    x = (rng.random((set_blocks, 1, BLOCK), dtype=np.float32) - 0.5).astype(np.float32)
    xin = torch.from_numpy(x).cuda()
    outs = torch.zeros(((warmup + steps) * set_blocks, 2, BLOCK), dtype=torch.float32, device="cuda")
    stride = set_blocks * 2 * BLOCK * 4

    def run(first, n):
        for s in range(first, first + n):
            rt.process_blocks(set_blocks, 2, out_ptr=outs.data_ptr() + s * stride, in_ptr=xin.data_ptr(), num_inputs=1)
    run(0, warmup)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(warmup, steps)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    total = (warmup + steps) * set_blocks
    us = 1e6 * dt / (steps * set_blocks)
    if gpu_only:      # (profiling passes: the GPU leg only, same launches)
        return {"config": "taps (GPU leg only)", "value": BLOCK / (us * 1e-6), "unit": "samples/s", "steps": steps, "ms_per_step": 1e3 * dt / steps, "us_per_block": us,
                "blocks_per_step": set_blocks, "launch_sets": steps + warmup}
    got = outs.cpu().numpy()
    cpu, kind = _ref_engine(sr)
    assert cpu.render(*_tap_graph())["result"] == 0
    ref = np.empty_like(got)
    lat = 0.0
    for b in range(total):
        xb = x[b % set_blocks]
        t1 = time.perf_counter()
        y = cpu.process(xb, 2, BLOCK)
        lat += time.perf_counter() - t1
        ref[b] = y
    cpu_us = 1e6 * lat / total
    plan, st = rt.describe_plan(), rt.stats()
    # fanIn + outs per node: per loop tapIn 1, sdelay 2, mul 3, add 3, svf 4, tapOut 2, tanh 2, + 4 consts; in 1; 2 x (add 5, root 2); bus
    alg = (8 * (1 + 2 + 3 + 3 + 4 + 2 + 2 + 4) + 1 + 2 * (5 + 2)) * BLOCK * 4 + 2 * BLOCK * 4
    return {
        "config": "feedback taps: 8 loops tapOut(lowpass(add(in, 0.7 * sdelay(tapIn)))) -> tanh under two roots, sr 48000, blockSize 512 (not a BASELINE "
                  "configuration; the path north_star calls 'feedback edges resolved by block-serial launches')",
        "protocol": f"elemhip_process_blocks, one step = one launch set of {set_blocks} blocks; input and outputs resident in HBM",
        "value": BLOCK / (us * 1e-6), "unit": "samples/s", "steps": steps, "warmup": warmup, "ms_per_step": 1e3 * dt / steps, "us_per_block": us,
        "taps_in_sets": plan.get("taps_in_sets"), "spec_launches": st["spec_launches"], "batch_launches": st["batch_launches"],
        "roofline": _roofline(alg, us, "sum(fanIn + outs) x 2 KB per block over the timed region; a tap loop keeps ONE block in flight per island: bound by the serial block-to-block hand-over",
                              **dict(zip(("traffic", "traffic_source"), _traffic("taps", blocks_per_launch=set_blocks)))),
        "cpu_baseline": {"value": BLOCK / (cpu_us * 1e-6), "unit": "samples/s", "cores": 1, "kind": kind,
                         "sample": f"all {total} blocks of the same graph and input, one process() call per block", "us_per_block": cpu_us},
        "speedup_vs_cpu_baseline": cpu_us / us,
        "parity": _parity(got, ref, f"every block of the run ({total} blocks: warm-up + all timed sets) vs the {kind} engine"),
    }


# ------------------------------------------------------------------------------------------------------------------------------
# C5 — BASELINE configs[4]: a live 128-voice graph under a mutation stream
# ------------------------------------------------------------------------------------------------------------------------------
def _extra_stage(y, k):
    """The seeded family of the shape-churn stream: an extra node chain that makes voice `k`'s island structurally new."""
    from elementary_amd import el
    ops = [lambda s: el.tanh(s), lambda s: el.mul(0.97, s), lambda s: el.add(0.001, s), lambda s: el.abs(s), lambda s: el.sin(s),
           lambda s: el.pole(0.5, s), lambda s: el.z(s), lambda s: el.max(s, -0.5), lambda s: el.min(s, 0.5), lambda s: el.sub(s, 0.002)]
    n = len(ops)
    code = 1 + k                                    # base-n digits of `code`: a distinct op sequence per k
    while code:
        y = ops[code % n](y)
        code //= n
    return y


def _c5_batches(voices, count, churn=False):
    """Instruction batches generated ahead of every timed region (the JS frontend is not on the path): batch 0 mounts the
    graph, every later batch replaces one voice the way the reconciler does (new voice nodes, a new mix add and root per channel)."""
    from elementary_amd import el, graphs
    from elementary_amd.reconciler import Renderer, batch_to_json
    sent = []
    r = Renderer(lambda b: sent.append(b) or 0)
    gen = [0] * voices
    extra = [None] * voices

    def voice(s):
        v = graphs.c2_voice(s + voices * gen[s])
        return _extra_stage(v, extra[s]) if extra[s] is not None else v

    def roots():
        vs = [voice(s) for s in range(voices)]
        return [el.add(*[vs[s] for s in range(voices) if s % 2 == ch]) for ch in range(2)]
    r.render(*roots())
    for b in range(count):
        s = (b * 37) % voices
        gen[s] += 1
        if churn:
            extra[s] = b
        r.render(*roots())
    return [batch_to_json(b) for b in sent], [sum(1 for i in b if i[0] == 0) for b in sent], [len(b) for b in sent]


def _c5_counted(rt, texts, commits, blocks_after, rate, keep=True, gc_every=16, process=None):
    """The CHECKABLE leg: commit k, then the first block of the new graph (timed with the commit), then `blocks_after` - 1 more
    synchronous blocks; commits paced at `rate` per second of wall clock when the engine is faster than that. Every rendered block
    is kept, so the reference engine driven through the same schedule must produce the same samples."""
    import gc as _pygc
    import numpy as np
    process = process or (lambda: rt.process(None, 2, BLOCK))
    assert rt.apply_instructions_json(texts[0]) == 0
    # (r06: the harness's own cyclic garbage collector stays out of the latency samples — for the reference engine's leg as well; a
    #  collection of this process's Python objects is a millisecond that neither engine spent)
    _pygc.collect()
    _gc_was = _pygc.isenabled()
    _pygc.disable()
    pre = 48
    out = np.empty((pre + commits * blocks_after, 2, BLOCK), dtype=np.float32) if keep else None
    n = 0
    for _ in range(pre):                         # root fade-in settles (20 ms = 2 blocks), envelopes open
        y = process()
        if keep:
            out[n] = y
        n += 1
    lat, lat_commit, lat_block, gcs = [], [], [], []
    t0 = time.perf_counter()
    for k in range(1, commits + 1):
        if rate:
            due = t0 + (k - 1) / rate
            while time.perf_counter() < due:
                time.sleep(0.0005)
        ta = time.perf_counter()
        assert rt.apply_instructions_json(texts[k]) == 0
        tb = time.perf_counter()
        y = process()
        tc = time.perf_counter()
        lat.append(1e3 * (tc - ta)); lat_commit.append(1e3 * (tb - ta)); lat_block.append(1e3 * (tc - tb))
        if keep:
            out[n] = y
        n += 1
        for _ in range(blocks_after - 1):
            y = process()
            if keep:
                out[n] = y
            n += 1
        if k % gc_every == 0:
            gcs.append(len(rt.gc()))
    wall = time.perf_counter() - t0
    if _gc_was:
        _pygc.enable()
    return {"commit_to_first_block_ms": {"p50": _pct(lat, 0.5), "p99": _pct(lat, 0.99), "max": max(lat), "count": len(lat)},
            "commit_call_ms": {"p50": _pct(lat_commit, 0.5), "p99": _pct(lat_commit, 0.99)},
            "first_block_ms": {"p50": _pct(lat_block, 0.5), "p99": _pct(lat_block, 0.99)},
            "wall_s": wall, "nodes_collected_per_gc": (sum(gcs) / len(gcs)) if gcs else None}, out


def _c5_free_running(rt, texts, render_chunk, seconds, rate, first=1):
    """The THROUGHPUT leg (the r02-r04 protocol): the graph renders as fast as it can in 64-block calls while the mutation stream,
    paced by the wall clock, replaces a voice `rate` times per second; gc every 16 batches."""
    t0 = time.perf_counter()
    n = 0
    while time.perf_counter() - t0 < 1.0:
        render_chunk(64); n += 64
    static = n * BLOCK / (time.perf_counter() - t0)
    applied, frames = first - 1, 0
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        due = first - 1 + int((time.perf_counter() - t0) * rate)
        if applied < min(due, len(texts) - 1):
            applied += 1
            assert rt.apply_instructions_json(texts[applied]) == 0
            render_chunk(1)
            frames += BLOCK
            if applied % 16 == 0:
                rt.gc()
        else:
            render_chunk(64); frames += 64 * BLOCK
    dt = time.perf_counter() - t0
    return {"static_samples_per_s": static, "mutating_samples_per_s": frames / dt, "ratio": frames / dt / static, "batches_applied": applied - first + 1, "seconds": dt}


def _c5_reference_worker(args):
    """The reference engine through the counted schedule, in a process of its own (so that it can run beside the GPU leg)."""
    texts, commits, blocks_after = args
    from elementary_amd import graphs
    cpu, kind = _ref_engine(graphs.C2_SAMPLE_RATE)
    stats, out = _c5_counted(cpu, texts, commits, blocks_after, rate=0.0)
    return stats, out, kind


def c5(commits: int = 520, blocks_after: int = 8, rate: float = 28.0, churn: bool = False, seconds: float = 6.0, gpu_only: bool = False):
    import multiprocessing as mp
    import numpy as np
    import torch
    from elementary_amd import graphs
    from elementary_amd.runtime import Runtime
    voices = 128
    free_batches = int(seconds * rate) + 8
    texts, creates, sizes = _c5_batches(voices, commits + free_batches, churn=churn)
    rt = Runtime(graphs.C2_SAMPLE_RATE, BLOCK, device=0)
    rt.set_option("specialize", 1)               # a live graph never waits for a compiler: background compilation, the product default
    counted, got = _c5_counted(rt, texts, commits, blocks_after, rate)
    st_mid = rt.stats()
    plan_mid = rt.describe_plan()
    out = torch.empty((64, 2, BLOCK), dtype=torch.float32, device="cuda")
    free = _c5_free_running(rt, texts, lambda k: rt.process_blocks(k, 2, out_ptr=out.data_ptr()), seconds, rate, first=commits + 1)
    if gpu_only:      # (profiling passes: the GPU legs only, same launches)
        return {"config": "C5 (GPU legs only)", "value": free["mutating_samples_per_s"], "unit": "samples/s", "free_running": free, "counted": counted,
                "blocks_rendered": rt.stats()["blocks_rendered"]}
    st = rt.stats()
    plan = rt.describe_plan()
    # the reference engine through the same schedule AFTER the GPU legs (a latency measurement: nothing else of this script runs
    # beside either engine), in a process of its own
    ctx = mp.get_context("spawn")
    with ctx.Pool(1) as pool:
        ref_stats, ref_out, kind = pool.apply(_c5_reference_worker, ((texts[:commits + 1], commits, blocks_after),))
    parity = _parity(got, ref_out, f"every block of the counted leg ({got.shape[0]} blocks: 48 settling + {commits} commits x {blocks_after} synchronous blocks, gc every 16 commits) "
                                   f"vs the {kind} engine driven through the same instruction batches on the same block schedule")
    # where the samples first differ, if they do (a diagnostic worth more than a bare `false`)
    if not parity["ok"]:
        bad = np.nonzero(np.abs(got.astype(np.float64) - ref_out).max(axis=(1, 2)) > parity["tolerance"])[0]
        parity["first_bad_block"] = int(bad[0]) if len(bad) else None
        parity["bad_blocks"] = int(len(bad))
    alg = graphs.c2_algorithmic_bytes(voices, 2, BLOCK)
    us_mut = 1e6 * BLOCK / free["mutating_samples_per_s"]
    ref_us_block = 1e6 * ref_stats["wall_s"] / (commits * blocks_after)
    jit = plan.get("jit", {})
    rec = {
        "config": ("C5 shape churn (VERDICT r04 #2d): as C5, but every replacement voice is STRUCTURALLY NEW (an extra op chain drawn from a seeded family behind the voice), "
                   "so every commit meets an island shape no kernel exists for" if churn else
                   "BASELINE configs[4] (C5): dynamic graph, 128 live C2 voices (~2060 nodes), one voice replaced per batch "
                   "(~18 CREATE_NODE, ~150 APPEND_CHILD, ACTIVATE_ROOTS, COMMIT), gc every 16 batches")
                  + f"; {rate:g} batches per wall-clock second; specialize = 1 (background compilation: the product default)",
        "protocol": f"counted leg: {commits} commits, each followed by {blocks_after} synchronous elemhip_process blocks (latency = apply_instructions + the "
                    f"first block of the new graph), every block checked; free-running leg: {seconds:g} s of 64-block elemhip_process_blocks calls under the same stream",
        "value": free["mutating_samples_per_s"], "unit": "samples/s", "steps": free["batches_applied"], "ms_per_step": 1e3 / rate,
        "free_running": free, "counted": counted,
        "commit_to_first_block_ms_p50": counted["commit_to_first_block_ms"]["p50"], "commit_to_first_block_ms_p99": counted["commit_to_first_block_ms"]["p99"],
        "instructions_per_batch": sum(sizes[1:]) / max(1, len(sizes) - 1), "nodes_created_per_batch": sum(creates[1:]) / max(1, len(creates) - 1),
        "plans_built": st["plans_built"], "plan_build_us_last": plan.get("build_us"),
        "kernels": {"spec_shapes_now": st["spec_shapes"], "spec_launches": st["spec_launches"], "jit": jit,
                    "interpreter_block_fraction_counted_leg": plan_mid.get("interp_block_fraction"),
                    "interpreter_block_fraction_whole_run": plan.get("interp_block_fraction"),
                    "spec_launches_counted_leg": st_mid["spec_launches"]},
        "roofline": _roofline(alg, us_mut, "algorithmic bytes of the 128-voice graph per block / block time of the free-running leg under mutation",
                              **dict(zip(("traffic", "traffic_source"), _traffic("c5_churn" if churn else "c5")))),
        "cpu_baseline": {"value": BLOCK / (ref_us_block * 1e-6), "unit": "samples/s", "cores": 1, "kind": kind,
                         "sample": f"the counted leg on the reference engine: {commits} commits x {blocks_after} blocks, unpaced (it cannot keep up with {rate:g} commits/s + rendering in real time "
                                   "at this size: commit + blocks take longer than the pacing interval)" if ref_stats["wall_s"] > commits / rate else
                                   f"the counted leg on the reference engine: {commits} commits x {blocks_after} blocks, unpaced",
                         "commit_to_first_block_ms": ref_stats["commit_to_first_block_ms"], "commit_call_ms": ref_stats["commit_call_ms"], "us_per_block_incl_commits": ref_us_block},
        "speedup_vs_cpu_baseline": free["mutating_samples_per_s"] / (BLOCK / (ref_us_block * 1e-6)),
        "latency_vs_cpu_baseline": ref_stats["commit_to_first_block_ms"]["p50"] / counted["commit_to_first_block_ms"]["p50"],
        "parity": parity,
    }
    return rec


def main():
    import torch   # before libelemhip.so: both bind the HIP runtime torch ships
    assert torch.cuda.is_available(), "needs a GPU: the HIP engine has no CPU fallback"
    torch.cuda.init()
    name = sys.argv[1]
    go = "--gpu-only" in sys.argv
    fn = {"c1": c1, "c3": (lambda: c3(gpu_only=go)), "taps": (lambda: taps(gpu_only=go)), "c5": (lambda: c5(commits=160, seconds=3.0, gpu_only=True) if go else c5()),
          "c5_churn": lambda: c5(commits=160, churn=True, seconds=4.0)}[name]
    print(json.dumps(fn()), flush=True)


if __name__ == "__main__":
    main()
