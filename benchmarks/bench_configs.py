#!/usr/bin/env python
"""Secondary configurations of BASELINE.json (C1, C3, C4, C5) on ONE GPU, each with the CPU engine
timed beside it on one host core.  bench.py (C2) is the headline number; this script produces the
other rows of DESIGN.md's measurement table.  One JSON line per configuration.

    python benchmarks/bench_configs.py [c1] [c3] [c4] [c5] [--blocks N]
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
BLOCK = 512


def _cpu_engine(sr, use_ref=True):
    import oracle
    if use_ref and oracle.have_ref():
        return oracle.RefRuntime(sr, BLOCK, bench_build=os.path.exists(oracle.REF_BENCH_SO)), "reference"
    return oracle.PortRuntime(sr, BLOCK), "port"


def _time_cpu(rt, n_in, n_out, x=None, budget=6.0, warm=8):
    import numpy as np
    xin = None if n_in == 0 else (x if x is not None else np.zeros((n_in, BLOCK), np.float32))
    for _ in range(warm):
        rt.process(xin, n_out, BLOCK)
    t0 = time.perf_counter()
    for _ in range(20):
        rt.process(xin, n_out, BLOCK)
    per = (time.perf_counter() - t0) / 20
    m = int(max(50, min(5000, budget / per)))
    t0 = time.perf_counter()
    for _ in range(m):
        rt.process(xin, n_out, BLOCK)
    return (time.perf_counter() - t0) / m, m


def _time_gpu(rt, n_in, n_out, blocks, xin=None, chunk=256):
    import torch
    out = torch.empty((min(blocks, chunk), n_out, BLOCK), dtype=torch.float32, device="cuda")
    chunk = out.shape[0]

    def run(total):
        done = 0
        while done < total:
            c = min(chunk, total - done)
            rt.process_blocks(c, n_out, out_ptr=out.data_ptr(), in_ptr=(xin.data_ptr() if xin is not None else 0), num_inputs=n_in)
            done += c
    run(chunk)
    torch.cuda.synchronize()
    rt.set_option("profile_launches", 1)
    t0 = time.perf_counter()
    run(blocks)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / blocks
    prof = rt.launch_profile()
    rt.set_option("profile_launches", 0)
    _time_gpu.last_profile = {"level_us_per_block": [1e3 * x / max(1, prof["blocks"]) for x in prof["level_ms"]],
                              "epilogue_us_per_block": 1e3 * prof["epilogue_ms"] / max(1, prof["blocks"]), "blocks": prof["blocks"]}
    return dt


def _native_host(roots, sr, blocks=4000, env=None):
    """The reference's cli/Benchmark.cpp protocol on a native host: examples/bench_cli (C++ over the facade, no Python, no HIP on
    its side of the C-ABI) renders `roots` with one synchronous process() call per 512-frame block."""
    import json
    import subprocess
    import tempfile
    from elementary_amd.reconciler import Renderer, batch_to_json
    exe = os.path.join(ROOT, "examples", "bench_cli")
    if not os.path.exists(exe):
        return None
    sent = []
    Renderer(lambda b: sent.append(b) or 0).render(*roots)
    with tempfile.TemporaryDirectory() as d:
        bpath = os.path.join(d, "batch.json")
        open(bpath, "w").write(batch_to_json(sent[0]))
        res = subprocess.run([exe, bpath, str(blocks), str(sr)], capture_output=True, text=True, timeout=300, env=dict(os.environ, **(env or {})))
    if res.returncode != 0:
        return {"error": res.stderr[-300:]}
    line = [l for l in res.stderr.splitlines() if l.startswith("{")]
    return json.loads(line[-1]) if line else None


def c1(args):
    from elementary_amd import graphs
    from elementary_amd.runtime import Runtime
    rt = Runtime(44100.0, BLOCK, device=0)
    rt.set_option("specialize", args.specialize)
    assert rt.render(*graphs.c1_graph())["result"] == 0
    g = _time_gpu(rt, 0, 2, args.blocks)
    cpu, kind = _cpu_engine(44100.0)
    assert cpu.render(*graphs.c1_graph())["result"] == 0
    c, m = _time_cpu(cpu, 0, 2)
    lv = rt.time_launches(2, 200)
    return {"config": "C1 cli/Benchmark graph (18 nodes, sr 44100)", "gpu_us_per_block": 1e6 * g, "gpu_samples_per_s": BLOCK / g,
            "cpu_us_per_block": 1e6 * c, "cpu_kind": kind, "cpu_blocks_timed": m, "speedup": c / g,
            "launch_us": [1e3 * v for v in lv], "note": "latency-bound: one serial phasor -> svf chain per channel",
            "native_host_process_call": _native_host(graphs.c1_graph(), 44100.0, env={"ELEMHIP_SPECIALIZE": "2"}),
            "native_host_process_call_c2_256_voices": _native_host(graphs.c2_graph(), graphs.C2_SAMPLE_RATE, env={"ELEMHIP_SPECIALIZE": "2"})}


def c3(args):
    import numpy as np
    import torch
    from elementary_amd import graphs
    from elementary_amd.runtime import Runtime
    ch = graphs.C3_CHANNELS
    rt = Runtime(graphs.C3_SAMPLE_RATE, BLOCK, device=0)
    irs = [graphs.c3_impulse_response(c) for c in range(ch)]
    for c in range(ch):
        assert rt.add_shared_resource(f"ir{c}", irs[c])
    assert rt.render(*graphs.c3_graph(ch))["result"] == 0
    set_blocks = args.batch_blocks or 1024
    rt.set_option("batch_blocks", set_blocks)
    for kv in args.opt:
        k_, v_ = kv.split("=", 1)
        rt.set_option(k_, float(v_))
    x = graphs.c3_input(ch, 64 * BLOCK)
    xin = torch.from_numpy(np.ascontiguousarray(x.reshape(ch, 64, BLOCK).transpose(1, 0, 2))).cuda().repeat(max(4, set_blocks // 64), 1, 1).contiguous()
    g = _time_gpu(rt, ch, ch, args.blocks, xin, chunk=max(256, set_blocks))
    lv = rt.time_launches(ch, 200)
    cpu, kind = _cpu_engine(graphs.C3_SAMPLE_RATE, use_ref=False)   # convolve exists only in the restatement (and the wasm build)
    for c in range(ch):
        assert cpu.add_shared_resource(f"ir{c}", irs[c])
    assert cpu.render(*graphs.c3_graph(ch))["result"] == 0
    c, m = _time_cpu(cpu, ch, ch, x[:, :BLOCK].copy())
    alg = graphs.c3_algorithmic_bytes(ch)
    conv_us = 1e3 * (lv[1] if len(lv) > 2 else lv[0])    # the convolve launch (level 0 once `in` / root are folded into it)
    # the reference's own engine (its prebuilt wasm build under Node): timed HERE, beside the GPU, when node and /root/reference
    # exist on this machine (benchmarks/c3_wasm_baseline.js); otherwise the figure recorded in the authoring container, labelled
    wasm = None
    import json as _json, os as _os, shutil as _sh, subprocess as _sp
    here = _os.path.dirname(_os.path.abspath(__file__))
    if _sh.which("node") and _os.path.isdir("/root/reference/js/packages"):
        try:
            r = _sp.run(["node", _os.path.join(here, "c3_wasm_baseline.js")], capture_output=True, text=True, timeout=300)
            w = _json.loads(r.stdout.strip().splitlines()[-1])
            wasm = {k: w[k] for k in ("us_per_block_mean", "us_per_block_p50", "us_per_block_p99", "host_cpu") if k in w}
            wasm["measured"] = "on this machine, in this run"
        except Exception as e_:
            wasm = {"error": str(e_)[:200]}
    if wasm is None or "error" in wasm:
        try:
            w = _json.load(open(_os.path.join(here, "..", "profiles", "r02", "c3_wasm_reference_cpu.json")))
            wasm = {k: w[k] for k in ("us_per_block_mean", "us_per_block_p50", "us_per_block_p99", "host_cpu", "measured_on")}
            wasm["measured"] = "NOT in this run: recorded in the authoring container (no node / no /root/reference on this machine)"
        except Exception:
            pass
    return {"config": "C3 8-channel convolution reverb, 96 000-tap IRs, sr 48000", "gpu_us_per_block": 1e6 * g,
            "gpu_path": f"elemhip_process_blocks: multi-block convolve kernels (fft / mac / ifft / finish per {set_blocks}-block launch set)",
            "gpu_launch_set_profile": getattr(_time_gpu, "last_profile", None), "batch_launches": rt.stats()["batch_launches"],
            "single_block_launch_note": "launch_us / conv_kernel_us below time the block-at-a-time path (elemhip_process)",
            "reference_wasm_cpu": wasm,
            "gpu_samples_per_s": BLOCK / g, "cpu_us_per_block": 1e6 * c, "cpu_kind": kind + " (plain radix-2 FFT, not Ooura)",
            "cpu_blocks_timed": m, "speedup": c / g, "launch_us": [1e3 * v for v in lv],
            "algorithmic_bytes_per_block": alg, "conv_kernel_us": conv_us,
            "conv_kernel_GBps_algorithmic": (alg / (conv_us * 1e-6) / 1e9) if conv_us else None,
            "conv_mfma": int(dict(kv.split("=", 1) for kv in args.opt).get("conv_mfma", 1)),
            "mac_flops_per_launch_set": 8.0 * ch * 188 * 512 * set_blocks}


def c4(args):
    from elementary_amd import graphs
    from elementary_amd.runtime import Runtime
    inst = args.instances
    rt = Runtime(graphs.C4_SAMPLE_RATE, BLOCK, device=0)
    rt.set_option("specialize", args.specialize)
    roots = [graphs.c4_instance(k) for k in range(inst)]
    t0 = time.perf_counter()
    assert rt.render(*roots)["result"] == 0
    build = time.perf_counter() - t0
    g = _time_gpu(rt, 0, inst, args.blocks)
    lv = rt.time_launches(inst, 100)
    cpu, kind = _cpu_engine(graphs.C4_SAMPLE_RATE)
    sub = min(inst, 16)
    assert cpu.render(*roots[:sub])["result"] == 0
    c, m = _time_cpu(cpu, 0, sub, budget=4.0)
    c_all = c * inst / sub
    return {"config": f"C4 {inst} independent offline render instances on one GPU (of 1024 over 8), sr 48000",
            "gpu_us_per_block_step": 1e6 * g, "gpu_instance_samples_per_s": inst * BLOCK / g,
            "cpu_us_per_block_step_1core": 1e6 * c_all, "cpu_kind": kind, "cpu_sample": f"{sub} instances x {m} blocks, scaled to {inst}",
            "speedup_vs_1core": c_all / g, "launch_us": [1e3 * v for v in lv], "plan_build_ms": 1e3 * build, "specialize": args.specialize,
            "launch_profile": getattr(_time_gpu, "last_profile", None),
            "spec_launches": rt.stats()["spec_launches"],
            "algorithmic_bytes_per_block_step": graphs.c4_algorithmic_bytes(inst)}


def _c5_batches(voices, count):
    """The instruction stream of config 5, generated ahead of the timed region (the reference's JS frontend is not part of
    the path being measured): batch 0 mounts the 128-voice graph, every later batch replaces one voice the way the
    reconciler does (new voice nodes, a new mix add and a new root per channel). Returns JSON texts + CREATE_NODE counts."""
    from elementary_amd import el, graphs
    from elementary_amd.reconciler import Renderer, batch_to_json
    sent = []
    r = Renderer(lambda b: sent.append(b) or 0)
    gen = [0] * voices

    def roots():
        vs = [graphs.c2_voice(s + voices * gen[s]) for s in range(voices)]
        return [el.add(*[vs[s] for s in range(voices) if s % 2 == ch]) for ch in range(2)]
    r.render(*roots())
    for b in range(count):
        gen[(b * 37) % voices] += 1
        r.render(*roots())
    return [batch_to_json(b) for b in sent], [sum(1 for i in b if i[0] == 0) for b in sent], [len(b) for b in sent]


def c5(args):
    """Dynamic graph (BASELINE configs[4]): a live 128-voice graph (~2060 nodes) renders as fast as it can
    while a mutation stream PACED BY THE WALL CLOCK replaces one voice per batch, 28 batches per second (~1000 node adds
    + as many removals per second); gc() every 16 batches. Reported: frames/s static and under mutation, commit -> first
    block of the new graph (apply_instructions + the first block), plan rebuild, hipGraph re-capture; the reference
    engine beside it under the same protocol."""
    import torch
    from elementary_amd import graphs
    from elementary_amd.runtime import Runtime
    voices, rate = 128, 28.0
    texts, creates, sizes = _c5_batches(voices, args.batches)
    out = torch.empty((64, 2, BLOCK), dtype=torch.float32, device="cuda")

    def drive(rt, render_chunk, seconds):
        assert rt.apply_instructions_json(texts[0]) == 0
        for _ in range(8):
            render_chunk(64)                       # settle the root fades, warm the plan
        t0 = time.perf_counter()
        n = 0
        while time.perf_counter() - t0 < 1.0:
            render_chunk(64); n += 64
        static = n * BLOCK / (time.perf_counter() - t0)
        lat, lat_commit, lat_block, applied, frames = [], [], [], 0, 0
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < seconds:
            due = int((time.perf_counter() - t0) * rate)
            if applied < min(due, len(texts) - 1):
                applied += 1
                ta = time.perf_counter()
                assert rt.apply_instructions_json(texts[applied]) == 0
                tb = time.perf_counter()
                render_chunk(1)                    # the first block rendered by the new render sequence
                tc = time.perf_counter()
                lat.append(1e3 * (tc - ta)); lat_commit.append(1e3 * (tb - ta)); lat_block.append(1e3 * (tc - tb))
                frames += BLOCK
                if applied % 16 == 0:
                    rt.gc()
            else:
                render_chunk(64); frames += 64 * BLOCK
        dt = time.perf_counter() - t0
        lat.sort(); lat_commit.sort(); lat_block.sort()
        pct = lambda a, q: a[min(len(a) - 1, int(q * len(a)))] if a else None
        return {"commit_call_ms_p50": pct(lat_commit, 0.5), "commit_call_ms_p99": pct(lat_commit, 0.99),
                "first_block_ms_p50": pct(lat_block, 0.5), "first_block_ms_p99": pct(lat_block, 0.99),
                "static_frames_per_s": static, "mutating_frames_per_s": frames / dt, "ratio": frames / dt / static, "batches_applied": applied,
                "commit_to_first_block_ms_p50": pct(lat, 0.5), "commit_to_first_block_ms_p99": pct(lat, 0.99)}

    def drive_two_threads(rt, seconds):
        """The realtime shape of config 5: a render thread calls process() block after block (synchronous, through host
        buffers) while a control thread applies the mutation stream at `rate` batches per second. Reported: the render
        call's latency distribution without and with the commits running (the engine plans outside the render lock, so the
        two should agree), and how long a commit call takes on the control thread."""
        import threading
        assert rt.apply_instructions_json(texts[0]) == 0
        for _ in range(64):
            rt.process(None, 2, BLOCK)

        def render_for(sec, stop=None):
            lat = []
            t_end = time.perf_counter() + sec
            while time.perf_counter() < t_end and not (stop and stop.is_set()):
                t0 = time.perf_counter()
                rt.process(None, 2, BLOCK)
                lat.append(1e6 * (time.perf_counter() - t0))
            lat.sort()
            return lat
        quiet = render_for(1.0)
        commits, done = [], threading.Event()

        def control():
            t0 = time.perf_counter()
            k = 0
            while time.perf_counter() - t0 < seconds and k < len(texts) - 1:
                due = int((time.perf_counter() - t0) * rate)
                if k < due:
                    k += 1
                    ta = time.perf_counter()
                    assert rt.apply_instructions_json(texts[k]) == 0
                    commits.append(1e3 * (time.perf_counter() - ta))
                    if k % 16 == 0:
                        rt.gc()
                else:
                    time.sleep(0.002)
            done.set()
        th = threading.Thread(target=control)
        import gc as _gc
        import sys as _sys
        _sys.setswitchinterval(1e-4)               # (CPython hands the GIL over every 5 ms by default: that, not the engine, would be the tail)
        _gc.collect(); _gc.disable()
        th.start()
        busy = render_for(seconds + 1.0, stop=done)
        th.join()
        _gc.enable()
        commits.sort()
        pct = lambda a, q: a[min(len(a) - 1, int(q * len(a)))] if a else None
        return {"render_call_us_quiet": {"p50": pct(quiet, 0.5), "p99": pct(quiet, 0.99), "max": quiet[-1] if quiet else None, "calls": len(quiet)},
                "render_call_us_while_committing": {"p50": pct(busy, 0.5), "p99": pct(busy, 0.99), "max": busy[-1] if busy else None, "calls": len(busy)},
                "commit_call_ms": {"p50": pct(commits, 0.5), "p99": pct(commits, 0.99), "count": len(commits)},
                "note": "render thread: synchronous elemhip_process of one 512-frame block (launch + D2H + sync); control thread: "
                        "apply_instructions at 28 batches per second + gc every 16; one Python process (the GIL is released inside both calls)"}

    rt = Runtime(graphs.C2_SAMPLE_RATE, BLOCK, device=0)
    rt.set_option("specialize", 1)   # a live graph never waits for a compiler: background mode (the product default)
    gpu = drive(rt, lambda k: rt.process_blocks(k, 2, out_ptr=out.data_ptr()), args.seconds)
    st = rt.stats()
    plan_info = rt.describe_plan()
    rt2 = Runtime(graphs.C2_SAMPLE_RATE, BLOCK, device=0)
    rt2.set_option("specialize", 1)
    threaded = drive_two_threads(rt2, min(args.seconds, 6.0))
    cpu, kind = _cpu_engine(graphs.C2_SAMPLE_RATE)
    ref = drive(cpu, lambda k: [cpu.process(None, 2, BLOCK) for _ in range(k)], min(args.seconds, 4.0))
    return {"config": "C5 dynamic graph: 128 live voices (~2060 nodes), one voice replaced per batch, 28 batches per wall-clock second, gc every 16",
            "instructions_per_batch": sum(sizes[1:]) / max(1, len(sizes) - 1), "nodes_created_per_batch": sum(creates[1:]) / max(1, len(creates) - 1),
            "node_adds_per_second": rate * sum(creates[1:]) / max(1, len(creates) - 1),
            "gpu": gpu, "gpu_two_threads": threaded, "plan_build_ms_last": st["last_plan_build_ms"], "plan_build_us_last": plan_info["build_us"],
            "island_programs_reused_per_replan": plan_info["plan_islands_reused"] / max(1, st["plans_built"] - 1), "program_heaps": plan_info["plan_prog_heaps"], "hipgraph_capture_ms_last": st["last_graph_capture_ms"],
            "hipgraph_captures": st["graph_captures"], "plans_built": st["plans_built"],
            "cpu_reference": ref, "cpu_kind": kind,
            "note": "instruction batches are generated before the timed region; commit -> first block = apply_instructions (graph "
                    "mutation + full plan build + upload) + one block through the per-block launch path while the roots cross-fade"}


def main():
    import torch   # before libelemhip.so: both must bind the same HIP runtime, torch's loads first
    assert torch.cuda.is_available(), "needs a GPU"
    torch.cuda.init()
    ap = argparse.ArgumentParser()
    ap.add_argument("configs", nargs="*", default=["c1", "c3", "c4", "c5"])
    ap.add_argument("--blocks", type=int, default=2048)
    ap.add_argument("--batch-blocks", type=int, default=0, help="blocks per launch set (0: c3 uses 1024, the others the engine default)")
    ap.add_argument("--instances", type=int, default=128)
    ap.add_argument("--batches", type=int, default=400)
    ap.add_argument("--seconds", type=float, default=8.0)
    ap.add_argument("--specialize", type=int, default=2, help="1 for c5 (background compilation, the product default) is set by c5 itself")
    ap.add_argument("--opt", action="append", default=[], metavar="KEY=VALUE", help="extra engine option (c3), repeatable")
    args = ap.parse_args()
    for name in args.configs:
        out = {"c1": c1, "c3": c3, "c4": c4, "c5": c5}[name](args)
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
