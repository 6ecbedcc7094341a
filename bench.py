#!/usr/bin/env python
"""bench.py — BASELINE.json's headline metric on its 1-GPU configuration (configs[1], "C2"):
audio samples/sec for the 4107-node / 256-voice subtractive-synth graph at blockSize 512.

A *step* is one pass of the hot path over one batch: ONE LAUNCH SET = 1024 consecutive 512-frame blocks of
that graph (524 288 output frames = 10.9 s of audio; --batch-blocks) rendered by the HIP engine through the offline entry point
(``elemhip_process_blocks``: one multi-block kernel launch per island level, outputs resident in HBM).
``--steps K --warmup W`` therefore renders W + K full launch sets whatever K is, and the roofline
figures are computed from the K timed steps themselves (wall clock for the headline fraction, HIP
event pairs recorded inside the timed region for the dominant kernel).  With N > 1 GPUs the voices
shard over the ranks with no data-path collective; the per-rank output buses are sum-reduced to
rank 0 over RCCL inside the timed region.  ``--scaling weak`` (default): every rank
renders its own 256-voice graph, ``value`` = frames of all N graphs / max-over-ranks wall time;
``--scaling strong``: the ONE 256-voice graph is split over the ranks, ``value`` = its frames / time.

Launch (N > 1):  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1
                 --master-port P bench.py --gpus N --steps K --warmup W
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0   # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec peak
BLOCK = 512


def cpu_baseline(target_seconds: float = 15.0):
    """Reference engine (oracle/_ref, -O3 -march=x86-64-v3 -ffp-contract=off) on ONE host core,
    same graph, cli/Benchmark.cpp protocol (warm-up then timed process() calls, steady clock)."""
    import numpy as np  # noqa: F401
    import oracle
    from elementary_amd import graphs

    if oracle.have_ref():
        rt = oracle.RefRuntime(graphs.C2_SAMPLE_RATE, BLOCK, bench_build=os.path.exists(oracle.REF_BENCH_SO))
        kind = "reference"
    elif oracle.have_port():
        rt = oracle.PortRuntime(graphs.C2_SAMPLE_RATE, BLOCK)
        kind = "port"
    else:
        return None
    assert rt.render(*graphs.c2_graph())["result"] == 0
    for _ in range(8):
        rt.process(None, 2, BLOCK)
    t0 = time.perf_counter()
    for _ in range(50):
        rt.process(None, 2, BLOCK)
    per = (time.perf_counter() - t0) / 50
    m = int(max(200, min(4000, target_seconds / per)))
    t0 = time.perf_counter()
    for _ in range(m):
        rt.process(None, 2, BLOCK)
    dt = time.perf_counter() - t0
    return {
        "value": BLOCK * m / dt, "unit": "samples/s", "cores": 1, "kind": kind,
        "sample": f"{m} blocks of {BLOCK} frames of the same 4107-node C2 graph, 1 thread, after 58 warm-up blocks",
        "ms_per_block": 1e3 * dt / m,
    }


def _mc_worker(args):
    first, count, blocks = args
    import oracle
    from elementary_amd import graphs
    rt = oracle.RefRuntime(graphs.C2_SAMPLE_RATE, BLOCK, bench_build=os.path.exists(oracle.REF_BENCH_SO))
    assert rt.render(*graphs.c2_graph(voices=count, channels=2, first_voice=first))["result"] == 0
    for _ in range(8):
        rt.process(None, 2, BLOCK)
    t0 = time.perf_counter()
    for _ in range(blocks):
        rt.process(None, 2, BLOCK)
    return time.perf_counter() - t0


def cpu_baseline_multicore(single_ms_per_block: float, target_seconds: float = 8.0):
    """SURVEY 8(d) fairness variant: the 256 voices partitioned over P reference Runtimes on P host cores
    (one process each, the host would sum P stereo buses per block: negligible, not timed)."""
    import multiprocessing as mp
    import oracle
    if not oracle.have_ref():
        return None
    cores = max(1, min(32, (os.cpu_count() or 1)))
    while 256 % cores:
        cores -= 1
    per = 256 // cores
    blocks = int(max(50, min(2000, target_seconds / (single_ms_per_block * 1e-3 * per / 256.0))))
    ctx = mp.get_context("spawn")   # the parent holds a HIP context: never fork it
    with ctx.Pool(cores) as pool:
        times = pool.map(_mc_worker, [(k * per, per, blocks) for k in range(cores)])
    dt = max(times)
    return {"value": BLOCK * blocks / dt, "unit": "samples/s", "cores": cores, "kind": "reference",
            "sample": f"{blocks} blocks, {cores} processes x {per} voices each (same 256-voice graph partitioned by voice)",
            "ms_per_block": 1e3 * dt / blocks}


def main_c4(args) -> None:
    """BASELINE configs[3] (C4): `--instances` independent offline render jobs per GPU (1024 over 8 GPUs = 128 per GPU).
    SURVEY.md 8(e): the unit of sharding is the whole render job, so ranks share nothing — no data-path collective, each
    job's output stays in its rank's HBM; weak scaling by construction. Same timing protocol as the headline workload."""
    import torch
    import torch.distributed as dist

    from elementary_amd import graphs
    from elementary_amd.runtime import Runtime

    world, rank, local = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} needs WORLD_SIZE={args.gpus} (launch with torch.distributed.run); got {world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP engine has no CPU fallback")
    torch.cuda.set_device(local)
    if world > 1:
        dist.init_process_group("nccl", rank=rank, world_size=world)
    inst, B = args.instances, max(1, min(1024, args.batch_blocks))
    rt = Runtime(graphs.C4_SAMPLE_RATE, BLOCK, device=local)
    rt.set_option("batch_blocks", B)
    rt.set_option("specialize", args.specialize)
    for kv in args.opt:
        k, v = kv.split("=", 1)
        rt.set_option(k, float(v))
    t0 = time.perf_counter()
    assert rt.render(*[graphs.c4_instance(inst * rank + k) for k in range(inst)])["result"] == 0
    build_ms = 1e3 * (time.perf_counter() - t0)
    spc = max(1, args.steps_per_call)
    out = torch.zeros((spc * B, inst, BLOCK), dtype=torch.float32, device="cuda")

    def run(steps: int) -> None:
        done = 0
        while done < steps:
            c = min(spc, steps - done)
            rt.process_blocks(c * B, inst, out_ptr=out.data_ptr())
            done += c

    run(args.warmup)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    rt.set_option("profile_launches", 1)
    t0 = time.perf_counter()
    run(args.steps)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    prof = rt.launch_profile()
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    if rank == 0:
        blocks = args.steps * B
        alg = graphs.c4_algorithmic_bytes(inst)                  # per block-step of one rank
        us = 1e6 * dt / blocks
        sets = max(1, prof["launch_sets"])
        stats = rt.stats()
        print(json.dumps({
            "metric": "instance-samples/sec, independent offline render instances (BASELINE configs[3])",
            "value": world * inst * BLOCK * blocks / dt, "unit": "samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": f"BASELINE configs[3] (C4): {inst} independent render instances per GPU ({inst * world} in all), "
                                   "each cycle -> biquad -> tanh, sr 48000, blockSize 512; one step = one launch set of "
                                   f"{B} blocks of every instance", "instances_per_gpu": inst, "instances_total": inst * world,
                       "blocks_per_step": B, "islands": stats["num_islands"], "launch_levels": stats["num_levels"],
                       "collectives": "none (jobs are independent; outputs stay on their rank)"},
            "us_per_block_step": us, "plan_build_ms": build_ms,
            "roofline": {"bound": "hbm", "achieved": alg / (us * 1e-6) / 1e9, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": alg / (us * 1e-6) / 1e9 / HBM_PEAK_GBPS, "traffic": None,
                         "algorithmic_bytes_per_block_step": alg,
                         "launch_us_per_step": [1e3 * x / sets for x in prof["level_ms"]],
                         "note": "bound by the biquad recurrence of one instance per workgroup (a lone wave), not by bytes"},
        }))
    if world > 1:
        dist.destroy_process_group()


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", choices=["c2", "c4"], default="c2",
                    help="c2 = the headline 256-voice synth (BASELINE configs[1]); c4 = independent render instances (configs[3])")
    ap.add_argument("--instances", type=int, default=128, help="--workload c4: render instances per GPU")
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=64, help="timed steps; one step = one launch set of --batch-blocks blocks")
    ap.add_argument("--warmup", type=int, default=8, help="untimed steps")
    ap.add_argument("--batch-blocks", type=int, default=1024, help="512-frame blocks per step (= per multi-block launch set)")
    ap.add_argument("--steps-per-call", type=int, default=1, help="steps per elemhip_process_blocks call")
    ap.add_argument("--graph-blocks", type=int, default=8, help="blocks per captured hipGraph (per-block launch path)")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--voices", type=int, default=256)
    ap.add_argument("--opt", action="append", default=[], metavar="KEY=VALUE", help="extra engine option (tuning experiments), repeatable")
    ap.add_argument("--specialize", type=int, default=2, choices=[0, 1, 2],
                    help="0: interpreter island kernels only; 2: per-island-shape kernels compiled at plan build (kcache/ on disk)")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak",
                    help="N > 1: weak = every rank renders --voices voices; strong = the --voices-voice graph is split over the ranks")
    args = ap.parse_args()
    if args.workload == "c4":
        return main_c4(args)

    import torch
    import torch.distributed as dist

    from elementary_amd import graphs
    from elementary_amd.runtime import Runtime
    from elementary_amd.sharded import reduce_bus, shard_range

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} needs WORLD_SIZE={args.gpus} (launch with torch.distributed.run); got {world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP engine has no CPU fallback")
    torch.cuda.set_device(local)
    if world > 1:
        dist.init_process_group("nccl", rank=rank, world_size=world)   # "nccl" == RCCL on ROCm

    # ---- this rank's shard of the synth (independent voices: no data-path collective) ----
    if args.scaling == "strong":
        lo, hi = shard_range(args.voices, world, rank)
        first, my_voices = lo, hi - lo
        total_voices = args.voices
    else:
        first, my_voices = args.voices * rank, args.voices
        total_voices = args.voices * world
    B = max(1, min(1024, args.batch_blocks))        # blocks per step
    rt = Runtime(graphs.C2_SAMPLE_RATE, BLOCK, device=local)
    rt.set_option("use_graph", 0 if args.no_graph else 1)
    rt.set_option("graph_blocks", args.graph_blocks)
    rt.set_option("batch_blocks", B)
    rt.set_option("specialize", args.specialize)
    for kv in args.opt:
        k, v = kv.split("=", 1)
        rt.set_option(k, float(v))
    t0 = time.perf_counter()
    res = rt.render(*graphs.c2_graph(voices=my_voices, channels=2, first_voice=first))
    assert res["result"] == 0, res["result"]
    build_ms = 1e3 * (time.perf_counter() - t0)

    spc = max(1, args.steps_per_call)
    bufs = [torch.zeros((spc * B, 2, BLOCK), dtype=torch.float32, device="cuda") for _ in range(2)]

    def run(steps: int) -> None:
        done, k, works = 0, 0, []
        while done < steps:
            c = min(spc, steps - done)
            buf = bufs[k % 2]
            if world > 1 and len(works) >= 2:
                works.pop(0).wait()            # the buffer we are about to overwrite has been reduced
                torch.cuda.current_stream().synchronize()   # the engine renders on its own stream
            rt.process_blocks(c * B, 2, out_ptr=buf.data_ptr())
            if world > 1:
                works.append(reduce_bus(buf[:c * B], dst=0, async_op=True))
            done += c
            k += 1
        for w in works:
            w.wait()

    run(args.warmup)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    rt.set_option("profile_launches", 1)            # HIP event pair around every launch of the timed region
    t0 = time.perf_counter()
    run(args.steps)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    prof = rt.launch_profile()
    rt.set_option("profile_launches", 0)
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    if rank == 0:
        stats = rt.stats()
        blocks = args.steps * B                      # blocks of THE graph rendered in the timed region (per rank in weak mode)
        graph_frames = BLOCK * blocks                # output frames of one rank's graph
        # weak scaling: N independent `--voices`-voice graphs; strong: one graph, its voices split over the ranks
        value = (world if args.scaling == "weak" else 1) * graph_frames / dt
        # ---- roofline (SURVEY §8(d) algorithmic bytes; whole timed region AND the dominant kernel by HIP events) ----
        alg_bytes = graphs.c2_algorithmic_bytes(my_voices, 2, BLOCK)         # per block, this rank
        us_per_block = 1e6 * dt / blocks
        achieved = alg_bytes / (us_per_block * 1e-6) / 1e9                    # GB/s over the timed region
        sets = max(1, prof["launch_sets"])
        lvl_us = [1e3 * x / sets for x in prof["level_ms"]]                   # mean per launch (one launch = B blocks)
        dom = max(range(len(lvl_us)), key=lambda i: lvl_us[i]) if lvl_us else 0
        lvl_alg = graphs.c2_level_algorithmic_bytes(my_voices, 2, BLOCK)     # per block: [voices level, mixer level]
        dom_alg = lvl_alg[dom] if dom < len(lvl_alg) else alg_bytes
        dom_achieved = dom_alg * B / (lvl_us[dom] * 1e-6) / 1e9 if lvl_us and lvl_us[dom] > 0 else None
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            try:
                traffic = json.load(open(tpath)).get("c2_hbm_bytes_per_block")
                traffic = traffic * B if traffic else None            # per launch set, like `achieved`'s basis
            except Exception:
                traffic = None
        # ---- latency figures outside the timed region ----
        rt.set_option("time_batch", 1)
        lv1 = rt.time_launches(2, 100)
        for _ in range(20):
            rt.process(None, 2, BLOCK)
        t1 = time.perf_counter()
        for _ in range(200):
            rt.process(None, 2, BLOCK)
        sync_us = 1e6 * (time.perf_counter() - t1) / 200

        out = {
            "metric": "audio samples/sec (48kHz-equiv), 4107-node/256-voice synth graph, blockSize=512",
            "value": value,
            "unit": "samples/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * dt / args.steps,
            "higher_is_better": True,
            "scaling": args.scaling,
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": "BASELINE configs[1] (C2): 256-voice subtractive synth, 4107 nodes "
                            "(2 blepsaw, train gate, pole envelope, svf lowpass, tanh per voice; 2 mix adds, 2 roots), "
                            "sr 48000, blockSize 512, 0 in / 2 out"
                            + (", per GPU" if args.scaling == "weak" and world > 1 else ""),
                "step": f"one launch set = {B} consecutive 512-frame blocks of the whole graph ({B * BLOCK} output frames)",
                "blocks_per_step": B,
                "frames_per_step": B * BLOCK,
                "nodes_per_gpu": stats["num_nodes_in_plan"],
                "voices_per_gpu": my_voices,
                "voices_total": total_voices,
                "block_size": BLOCK,
                "mode": "offline elemhip_process_blocks, output bus resident in HBM"
                        + (", RCCL sum-reduce of the bus to rank 0 per call" if world > 1 else ""),
                "steps_per_call": spc,
                "pipelined_blocks_in_flight": rt.describe_plan()["islands"][0]["copies"],
                "islands": stats["num_islands"], "launch_levels": stats["num_levels"], "max_lds_bytes": stats["max_lds_bytes"],
                "island_kernels": ("run-time specialised per island shape (hiprtc, gfx950): %d shape(s) covering %d islands, %d launches; "
                                   "compile wait %.0f ms inside plan_build_ms (0 = on-disk cache hit)"
                                   % (stats["spec_shapes"], stats["spec_islands"], stats["spec_launches"], stats["last_jit_wait_ms"]))
                                  if args.specialize and stats["spec_launches"] else "ahead-of-time interpreter kernel",
            },
            "us_per_block": us_per_block,
            "realtime_factor_48k": value / 48000.0,
            "plan_build_ms": build_ms,
            "sync_process_us_per_block": sync_us,
            "roofline": {
                "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic,
                "basis": "timed region: algorithmic bytes per block x blocks / wall time of the K timed steps",
                "algorithmic_bytes_per_block": alg_bytes, "algorithmic_bytes_per_step": alg_bytes * B,
                "dominant_kernel": {
                    "kernel": "elemhip_spec_island" if (args.specialize and stats["spec_launches"]) else "elemhip_island_kernel", "level": dom, "blocks_per_launch": B,
                    "us_per_launch": lvl_us[dom] if lvl_us else None,
                    "algorithmic_bytes_per_launch": dom_alg * B,
                    "achieved": dom_achieved, "frac": (dom_achieved / HBM_PEAK_GBPS) if dom_achieved else None,
                    "timing": "HIP event pairs on the engine's stream around every launch of the timed region",
                },
                "launch_us_per_step": lvl_us, "epilogue_us_per_step": 1e3 * prof["epilogue_ms"] / sets,
                "launch_sets_profiled": prof["launch_sets"],
                "kernel_time_fraction_of_step": (sum(lvl_us) + 1e3 * prof["epilogue_ms"] / sets) / (1e3 * 1e3 * dt / args.steps) if lvl_us else None,
                "single_block_launch_us": [1e3 * x for x in lv1],
                "note": "`traffic` = PMC HBM bytes (2 x FETCH_SIZE + WRITE_SIZE, MI355X_MICROARCH.md) of one launch set; buffers "
                        "inside an island live in LDS and never reach HBM, so traffic sits far below the algorithmic bytes",
            },
        }
        if world == 1 and not args.no_cpu_baseline:
            cb = cpu_baseline()
            if cb:
                out["cpu_baseline"] = cb
                out["speedup_vs_cpu_baseline"] = out["value"] / cb["value"]
                try:
                    mc = cpu_baseline_multicore(cb["ms_per_block"])
                except Exception as e:   # the fairness variant must never cost the headline line
                    mc = {"error": repr(e)}
                if mc:
                    out["cpu_baseline_all_cores"] = mc
                    if "value" in mc:
                        out["speedup_vs_cpu_all_cores"] = out["value"] / mc["value"]
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
