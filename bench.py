#!/usr/bin/env python
"""bench.py — BASELINE.json's headline metric on its 1-GPU configuration (configs[1], "C2"):
audio samples/sec for the 4107-node / 256-voice subtractive-synth graph at blockSize 512.

A *step* is one 512-frame block of that graph rendered by the HIP engine through the offline
entry point (``elemhip_process_blocks``: outputs stay resident in HBM).  With N > 1 GPUs every rank
renders its own 256-voice shard (weak scaling, independent voices, no data-path collective) and
the per-rank output buses are sum-reduced to rank 0 over RCCL inside the timed region
(SURVEY.md §8(e)).  ``value`` = frames rendered by all ranks / max-over-ranks wall time.

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


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=4096)
    ap.add_argument("--warmup", type=int, default=256)
    ap.add_argument("--chunk", type=int, default=256, help="blocks per elemhip_process_blocks call")
    ap.add_argument("--graph-blocks", type=int, default=8, help="blocks per captured hipGraph (per-block launch path)")
    ap.add_argument("--batch-blocks", type=int, default=64, help="blocks per multi-block launch (1 = per-block launches)")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--voices", type=int, default=256)
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    from elementary_amd import graphs
    from elementary_amd.runtime import Runtime
    from elementary_amd.sharded import reduce_bus

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

    # ---- build this rank's shard: voices [256*rank, 256*rank + 256) of a 256*N-voice synth ----
    rt = Runtime(graphs.C2_SAMPLE_RATE, BLOCK, device=local)
    rt.set_option("use_graph", 0 if args.no_graph else 1)
    rt.set_option("graph_blocks", args.graph_blocks)
    rt.set_option("batch_blocks", args.batch_blocks)
    t0 = time.perf_counter()
    res = rt.render(*graphs.c2_graph(voices=args.voices, channels=2, first_voice=args.voices * rank))
    assert res["result"] == 0, res["result"]
    build_ms = 1e3 * (time.perf_counter() - t0)

    chunk = max(1, min(args.chunk, args.steps))
    bufs = [torch.zeros((chunk, 2, BLOCK), dtype=torch.float32, device="cuda") for _ in range(2)]

    def run(blocks: int) -> None:
        done, k, works = 0, 0, []
        while done < blocks:
            c = min(chunk, blocks - done)
            buf = bufs[k % 2]
            if world > 1 and len(works) >= 2:
                works.pop(0).wait()            # the buffer we are about to overwrite has been reduced
                torch.cuda.current_stream().synchronize()   # the engine renders on its own stream
            rt.process_blocks(c, 2, out_ptr=buf.data_ptr())
            if world > 1:
                works.append(reduce_bus(buf[:c], dst=0, async_op=True))
            done += c
            k += 1
        for w in works:
            w.wait()

    run(args.warmup)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(args.steps)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    if rank == 0:
        stats = rt.stats()
        # ---- roofline of the dominant kernel (elemhip_island_kernel), HIP events on the engine's stream ----
        # One launch of a level renders `batch` blocks (the same launches the timed region issued).
        rt.set_option("time_batch", args.batch_blocks)
        lv = rt.time_launches(2, 50)
        batch = rt.last_time_batch
        island_ms = sum(lv[:-1])
        alg_bytes = graphs.c2_algorithmic_bytes(args.voices, 2, BLOCK)
        achieved = alg_bytes * batch / (island_ms * 1e-3) / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                traffic = tj.get("c2_hbm_bytes_per_launch_set") if batch > 1 else tj.get("c2_hbm_bytes_per_block")
            except Exception:
                traffic = None
        rt.set_option("time_batch", 1)
        lv1 = rt.time_launches(2, 100)
        # ---- synchronous single-block latency through host buffers (cli/Benchmark.cpp style) ----
        for _ in range(20):
            rt.process(None, 2, BLOCK)
        t1 = time.perf_counter()
        for _ in range(200):
            rt.process(None, 2, BLOCK)
        sync_us = 1e6 * (time.perf_counter() - t1) / 200

        out = {
            "metric": "audio samples/sec (48kHz-equiv), 4107-node/256-voice synth graph per GPU, blockSize=512",
            "value": world * BLOCK * args.steps / dt,
            "unit": "samples/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * dt / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": "BASELINE configs[1] (C2): 256-voice subtractive synth, 4107 nodes "
                            "(2 blepsaw, train gate, pole envelope, svf lowpass, tanh per voice; 2 mix adds, 2 roots), "
                            "sr 48000, blockSize 512, 0 in / 2 out, per GPU",
                "nodes_per_gpu": stats["num_nodes_in_plan"],
                "voices_per_gpu": args.voices,
                "block_size": BLOCK,
                "mode": "offline elemhip_process_blocks, output bus resident in HBM"
                        + (", RCCL sum-reduce of the bus to rank 0 per chunk" if world > 1 else ""),
                "blocks_per_call": chunk,
                "blocks_per_launch": batch,
                "pipelined_blocks_in_flight": rt.describe_plan()["islands"][0]["copies"],
                "islands": stats["num_islands"], "launch_levels": stats["num_levels"], "max_lds_bytes": stats["max_lds_bytes"],
            },
            "realtime_factor_48k": world * BLOCK * args.steps / dt / 48000.0,
            "plan_build_ms": build_ms,
            "sync_process_us_per_block": sync_us,
            "roofline": {
                "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic,
                "kernel": "elemhip_island_kernel", "launches_per_batch": len(lv) - 1, "blocks_per_launch": batch,
                "kernel_us_per_launch": [1e3 * x for x in lv[:-1]], "epilogue_us": 1e3 * lv[-1],
                "kernel_us_per_block": [1e3 * x / batch for x in lv[:-1]],
                "event_pair_overhead_us_subtracted": 1e3 * rt.last_event_overhead_ms,
                "algorithmic_bytes_per_block": alg_bytes, "algorithmic_bytes_per_launch_set": alg_bytes * batch,
                "single_block_launch_us": [1e3 * x for x in lv1],
                "note": "achieved = SURVEY §8(d) algorithmic bytes of the blocks one launch set renders / summed HIP-event "
                        "duration of that set's island-kernel launches (one per level); `traffic` = PMC HBM bytes of the same "
                        "launch set; buffers inside an island live in LDS and never reach HBM",
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
