#!/usr/bin/env python
"""bench.py — BASELINE.json's headline metric on its 1-GPU configuration (configs[1], "C2"):
audio samples/sec for the 4107-node / 256-voice subtractive-synth graph at blockSize 512.

A *step* is one pass of the hot path over one batch: ONE LAUNCH SET = 1024 consecutive 512-frame blocks of
that graph (524 288 output frames = 10.9 s of audio; --batch-blocks) rendered by the HIP engine through the offline
entry point and DELIVERED TO HOST MEMORY, where the reference's own callers receive their samples
(``elemhip_process_blocks_host``: planar host arrays, launch sets staged through pinned double buffers, the D2H of
set k overlapped with the rendering of set k + 1).  ``--steps K --warmup W`` renders W + K full launch sets whatever K
is; the roofline figures are computed from the K timed steps themselves (wall clock for the headline fraction, HIP
event pairs recorded inside the timed region for the dominant kernel).  After the timed region the output is CHECKED:
the head of the stream against the reference engine on one host core, the last blocks of the last timed step against
the reference engine advanced to the same block on all host cores (``parity_max_abs_err``).  The device-resident
rate (``elemhip_process_blocks``, outputs left in HBM) is measured right after and reported as a sub-field.
With N > 1 GPUs the voices shard over the ranks with no data-path collective; the per-rank output buses are
sum-reduced to rank 0 over RCCL inside the timed region and rank 0 copies the reduced bus to pinned
host memory.  ``--scaling weak`` (default): every rank renders its own 256-voice graph, ``value`` = frames of all N
graphs / max-over-ranks wall time; ``--scaling strong``: the ONE 256-voice graph is split over the ranks.

Output (r06): every full record — the other configurations' (C1, C3, C4, C5, C5 shape churn, taps: a process each, after the
headline's timed region) and then the headline's own — is printed as a JSON line of its own, tagged ``"record": <name>``; the LAST
stdout line is ONE compact record under 4 KB (benchmarks/headline.py: metric / value / unit / n_gpus / steps / warmup / ms_per_step /
dtype / config / roofline / cpu_baseline / parity + a `configs` map of six numbers per configuration) — the line the driver parses.
``--workload c4`` (BASELINE configs[3]) times the device-resident render of the jobs (`value`); the delivery of every job's samples
into host arrays (bound by the host link) and the 256-job full-chip rate are sub-records.

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


def cpu_baseline(target_seconds: float = 15.0, keep_blocks: int = 0, voices: int = 256):
    """Reference engine (oracle/_ref, -O3 -march=x86-64-v3 -ffp-contract=off) on ONE host core,
    same graph, cli/Benchmark.cpp protocol (warm-up then timed process() calls, steady clock).
    The first `keep_blocks` blocks of its output stream are kept (``head``, [2, frames]) for the parity check."""
    import numpy as np
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
    assert rt.render(*graphs.c2_graph(voices=voices))["result"] == 0
    head = []

    def step():
        y = rt.process(None, 2, BLOCK)
        if len(head) < keep_blocks:
            head.append(y)

    for _ in range(8):
        step()
    t0 = time.perf_counter()
    for _ in range(50):
        step()
    per = (time.perf_counter() - t0) / 50
    m = int(max(200, min(4000, target_seconds / per)))
    t0 = time.perf_counter()
    for _ in range(m):
        step()
    dt = time.perf_counter() - t0
    return {
        "value": BLOCK * m / dt, "unit": "samples/s", "cores": 1, "kind": kind,
        "sample": f"{m} blocks of {BLOCK} frames of the same {voices}-voice C2 graph, 1 thread, after 58 warm-up blocks",
        "ms_per_block": 1e3 * dt / m,
        "head": np.concatenate(head, axis=1) if head else None,
    }


def _mc_worker(args):
    first, count, blocks, tail = args
    import numpy as np
    import oracle
    from elementary_amd import graphs
    rt = oracle.RefRuntime(graphs.C2_SAMPLE_RATE, BLOCK, bench_build=os.path.exists(oracle.REF_BENCH_SO))
    assert rt.render(*graphs.c2_graph(voices=count, channels=2, first_voice=first))["result"] == 0
    last = []
    t0 = time.perf_counter()
    for k in range(blocks):
        y = rt.process(None, 2, BLOCK)
        if k >= blocks - tail:
            last.append(y.astype(np.float64))
    return time.perf_counter() - t0, (np.concatenate(last, axis=1) if last else None)


def cpu_baseline_multicore(single_ms_per_block: float, target_seconds: float = 8.0, advance_blocks: int = 0, tail: int = 0,
                           budget_seconds: float = 45.0, voices: int = 256):
    """SURVEY 8(d) fairness variant: the 256 voices partitioned over P reference Runtimes on P host cores
    (one process each, the host would sum P stereo buses per block: negligible, not timed).
    With `advance_blocks` the P engines render exactly that many blocks FROM TIME ZERO (when the estimate fits
    `budget_seconds`) and the partial buses of the last `tail` blocks come back summed in voice order (float64):
    the reference's stream at the end of the benchmark's timed region, for the parity check."""
    import multiprocessing as mp
    import numpy as np
    import oracle
    if not oracle.have_ref():
        return None
    cores = max(1, min(32, (os.cpu_count() or 1)))
    while voices % cores:
        cores -= 1
    per = voices // cores
    per_block_s = single_ms_per_block * 1e-3 * per / float(voices)
    blocks = int(max(50, min(2000, target_seconds / per_block_s)))
    advanced = bool(advance_blocks) and advance_blocks * per_block_s * 1.3 <= budget_seconds
    if advanced:
        blocks = int(advance_blocks)
    ctx = mp.get_context("spawn")   # the parent holds a HIP context: never fork it
    with ctx.Pool(cores) as pool:
        res = pool.map(_mc_worker, [(k * per, per, blocks, tail if advanced else 0) for k in range(cores)])
    dt = max(r[0] for r in res)
    out = {"value": BLOCK * blocks / dt, "unit": "samples/s", "cores": cores, "kind": "reference",
           "sample": f"{blocks} blocks from time zero, {cores} processes x {per} voices each (same {voices}-voice graph partitioned by voice)",
           "ms_per_block": 1e3 * dt / blocks, "tail": None}
    if advanced and tail:
        acc = np.zeros_like(res[0][1])
        for r in res:               # voice order: process k owns voices [k * per, (k + 1) * per)
            acc += r[1]
        out["tail"] = acc
    return out


def _c4_worker(args):
    """One host process of the C4 CPU leg: ONE reference Runtime PER INSTANCE (each offline render job owns its
    engine, offline-renderer/index.ts:87-133), rendered job after job, `blocks` blocks each from time zero; the last `tail`
    blocks of every job come back."""
    ks, blocks, tail, core = args
    import numpy as np
    import oracle
    from elementary_amd import graphs
    try:
        os.sched_setaffinity(0, {core})
    except Exception:
        pass
    rts = []
    for k in ks:
        rt = oracle.RefRuntime(graphs.C4_SAMPLE_RATE, BLOCK, bench_build=os.path.exists(oracle.REF_BENCH_SO))
        assert rt.render(graphs.c4_instance(k))["result"] == 0
        rts.append(rt)
    tails = {}
    t0 = time.perf_counter()
    for k, rt in zip(ks, rts):
        last = []
        for b in range(blocks):
            y = rt.process(None, 1, BLOCK)
            if b >= blocks - tail:
                last.append(y[0].copy())
        if tail:
            tails[k] = np.concatenate(last)
    return time.perf_counter() - t0, tails


def c4_cpu_baseline(inst: int, first: int = 0, target_seconds: float = 12.0):
    """The reference engine on the host cores of this box, the way the offline renderer would run the same jobs: P pinned
    processes, the `inst` jobs dealt round-robin, one Runtime per job. A bounded sample: M blocks per job."""
    import multiprocessing as mp
    import oracle
    if not oracle.have_ref():
        return None
    avail = sorted(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else list(range(os.cpu_count() or 1))
    P = max(1, min(len(avail), inst))            # every host core this process may use, up to one per job (no 32-core cap)
    ctx = mp.get_context("spawn")       # the parent holds a HIP context: never fork it
    with ctx.Pool(P) as pool:
        cal = pool.map(_c4_worker, [([first], 400, 0, avail[0])])[0][0] / 400.0          # seconds per job-block on one core
        per_proc = (inst + P - 1) // P
        M = int(max(200, min(20000, target_seconds / (cal * per_proc))))
        res = pool.map(_c4_worker, [([first + k for k in range(p_, inst, P)], M, 0, avail[p_ % len(avail)]) for p_ in range(P)])
    dt = max(r[0] for r in res)
    return {"value": inst * BLOCK * M / dt, "unit": "samples/s", "cores": P, "kind": "reference",
            "sample": f"{M} blocks of {BLOCK} frames of each of the same {inst} render jobs from time zero, one reference Runtime per job, "
                      f"{P} pinned host processes ({per_proc} jobs each, job after job)",
            "us_per_job_block_one_core": 1e6 * cal, "ms_per_block_step": 1e3 * dt / M}


def c4_parity(timed_host, inst: int, first: int, total_blocks: int, tail: int = 64, jobs: int = 8, budget_seconds: float = 90.0,
              us_per_job_block: float = 15.0):
    """The LAST `tail` blocks of the timed region for `jobs` of the render jobs, against reference engines advanced from
    time zero through all `total_blocks` blocks (one process per job)."""
    import multiprocessing as mp
    import numpy as np
    import oracle
    if not oracle.have_ref() or timed_host is None:
        return None
    picks = sorted({k for k in (0, 1, 2, 3, inst // 2 - 1, inst // 2, inst - 2, inst - 1, 17, 42) if 0 <= k < inst})[:max(1, jobs)]
    if total_blocks * us_per_job_block * 1e-6 > budget_seconds:
        return {"ok": None, "note": f"not checked: advancing a reference engine through {total_blocks} blocks exceeds the {budget_seconds:.0f} s budget"}
    tail = min(tail, total_blocks)
    avail = sorted(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else list(range(os.cpu_count() or 1))
    ctx = mp.get_context("spawn")
    with ctx.Pool(min(len(picks), max(1, len(avail)))) as pool:
        res = pool.map(_c4_worker, [([first + k], total_blocks, tail, avail[i % len(avail)]) for i, k in enumerate(picks)])
    worst, peak = 0.0, 0.0
    for (dt_, tails), k in zip(res, picks):
        ref = tails[first + k]
        got = timed_host[k, -tail * BLOCK:]
        worst = max(worst, float(np.abs(got - ref).max()))
        peak = max(peak, float(np.abs(ref).max()))
    tol = 1e-6 * max(1.0, peak)
    return {"ok": bool(worst <= tol), "max_abs_err": worst, "tolerance": tol, "jobs_checked": [first + k for k in picks], "blocks_checked_per_job": tail,
            "reference_advanced_blocks": total_blocks, "max_abs_ref": peak,
            "what": f"last {tail} blocks of the timed region of {len(picks)} render jobs vs reference engines advanced from time zero through all {total_blocks} blocks"}


def _emit(full: dict) -> None:
    """Every full record on a JSON line of its own, then ONE compact line (< 4 KB: benchmarks/headline.py) LAST — the line the
    driver parses. The full records also go to gpurun_out/bench_records.jsonl (scratch)."""
    sys.path.insert(0, os.path.join(ROOT, "benchmarks"))
    import headline
    rpath = None
    try:
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        rpath = os.path.join(ROOT, "gpurun_out", "bench_records.jsonl")
    except OSError:
        pass
    headline.emit(full, records_path=rpath)


def run_configs(names, timeout_s: float):
    """The other configurations of BASELINE.json, driver-run (VERDICT r04 "next" #1): each in a process of its own AFTER the headline's
    timed region (this process's engine is idle meanwhile), each record with its own value / ms_per_step x steps / roofline /
    cpu_baseline / parity on the samples it timed (benchmarks/driver_configs.py; C4 is this script's own `--workload c4`). A
    configuration that fails or runs out of time costs its own sub-record, never the headline line."""
    import subprocess
    recs = {}
    t_all = time.perf_counter()
    for name in names:
        if name == "c4":
            cmd = [sys.executable, os.path.abspath(__file__), "--workload", "c4", "--steps", "64", "--warmup", "8"]
        else:
            cmd = [sys.executable, os.path.join(ROOT, "benchmarks", "driver_configs.py"), name]
        t0 = time.perf_counter()
        try:
            res = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout_s, cwd=ROOT)
            lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
            if res.returncode == 0 and lines:
                # `bench.py --workload c4` ends with its own compact line: the full record is the "headline_full" line before it
                full = [ln for ln in lines if ln.startswith('{"record": "headline_full"')]
                rec = json.loads(full[-1] if full else lines[-1])
                rec.pop("record", None)
            else:
                rec = {"error": f"exit code {res.returncode}", "stderr_tail": res.stderr[-600:]}
        except subprocess.TimeoutExpired:
            rec = {"error": f"no result within {timeout_s:.0f} s"}
        except Exception as e:      # noqa: BLE001
            rec = {"error": repr(e)[:300]}
        rec["wall_s"] = time.perf_counter() - t0
        recs[name] = rec
    recs["total_wall_s"] = time.perf_counter() - t_all
    recs["all_parity_ok"] = all(bool((recs[n].get("parity") or {}).get("ok")) for n in names)
    return recs


def main_c4(args) -> None:
    """BASELINE configs[3] (C4): `--instances` independent offline render jobs per GPU (1024 over 8 GPUs = 128 per GPU),
    "RCCL output gather". SURVEY.md 8(e): the unit of sharding is the whole render job, so ranks share nothing while they
    render; the one exchange is the gather of the per-job outputs. N = 1 (r06): the timed region renders into HBM (`value` is not
    a PCIe-inclusive rate); the delivery of every job's samples to host memory (elemhip_process_blocks_host, `--host-delivered`
    times THAT instead) and the 256-job rate follow as sub-records. N > 1: each rank renders into HBM and the outputs of every
    chunk are gathered on rank 0 over RCCL (sharded.gather_outputs) INSIDE the timed region. Same timing protocol as the headline."""
    import numpy as np
    import torch
    import torch.distributed as dist

    from elementary_amd import graphs
    from elementary_amd.runtime import Runtime
    from elementary_amd.sharded import gather_outputs

    world, rank, local = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and (args.gpus > 1 or world > 1):
        raise SystemExit(f"--gpus {args.gpus} needs WORLD_SIZE={args.gpus} (launch with torch.distributed.run); got {world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP engine has no CPU fallback")
    if args.shared_gpu:
        local = 0
    torch.cuda.set_device(local)
    ranks_seen = 1
    if world > 1:
        dist.init_process_group(args.backend, rank=rank, world_size=world)
        ranks_seen = dist.get_world_size()
        if ranks_seen != args.gpus:
            raise SystemExit(f"--gpus {args.gpus} but the process group has {ranks_seen} ranks")
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
    # r06: the timed region leaves the jobs' samples in HBM at every N (`value` is never a PCIe-inclusive rate: 128 jobs x 2 KB per
    # block-step are 268 MB per launch set, which the host link — not the kernels — bounds at ~11.7 ms); the delivery into the
    # caller's host arrays (elemhip_process_blocks_host, r03-r05's `value`) is measured right after and reported as `host_delivered`
    host_mode = world == 1 and args.host_delivered
    out = torch.zeros((spc * B, inst, BLOCK), dtype=torch.float32, device="cuda")
    gathered_elems = 0

    def run(steps: int, host_out=None) -> None:
        nonlocal gathered_elems
        if host_mode:
            rt.process_blocks_host(None, inst, steps * B * BLOCK, out=host_out)
            return
        done = 0
        while done < steps:
            c = min(spc, steps - done)
            rt.process_blocks(c * B, inst, out_ptr=out.data_ptr())
            if world > 1:          # [blocks, inst, frames] -> per-job outputs [inst, blocks, frames], gathered on rank 0
                g = gather_outputs(out[:c * B].transpose(0, 1).contiguous(), dst=0)
                if g is not None:
                    gathered_elems += g.numel()
            done += c

    warm_host = np.empty((inst, max(1, args.warmup) * B * BLOCK), dtype=np.float32) if host_mode else None
    timed_host = np.zeros((inst, args.steps * B * BLOCK), dtype=np.float32) if host_mode else None
    if args.warmup:
        run(args.warmup, warm_host)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    rt.set_option("profile_launches", 1)
    gathered_elems = 0
    t0 = time.perf_counter()
    run(args.steps, timed_host)
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
        total_blocks = (args.warmup + args.steps) * B
        if host_mode:
            parity = c4_parity(timed_host, inst, inst * rank, total_blocks)
        elif world == 1:
            # the last timed step is still in the device buffer [block][job][frame]: its last blocks, job by job, against reference engines
            # advanced through the whole run
            tail_blocks = min(64, B)
            last = out[((args.steps - 1) % spc) * B:((args.steps - 1) % spc + 1) * B][-tail_blocks:].permute(1, 0, 2).reshape(inst, tail_blocks * BLOCK).cpu().numpy()
            parity = c4_parity(last, inst, inst * rank, total_blocks, tail=tail_blocks)
        else:
            parity = None
        host_delivered = None
        if world == 1 and not host_mode and not args.no_host_leg:
            hs = max(2, min(args.steps, 8))
            hbuf = np.zeros((inst, hs * B * BLOCK), dtype=np.float32)
            rt.process_blocks_host(None, inst, B * BLOCK, out=hbuf[:, :B * BLOCK])      # staging buffers, pages
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            rt.process_blocks_host(None, inst, hs * B * BLOCK, out=hbuf)
            dth = time.perf_counter() - t1
            host_delivered = {"value": inst * BLOCK * hs * B / dth, "unit": "samples/s", "ms_per_step": 1e3 * dth / hs, "steps": hs,
                              "bytes_per_step": inst * B * BLOCK * 4, "delivered_GBps": inst * B * BLOCK * 4 * hs / dth / 1e9,
                              "mode": "elemhip_process_blocks_host: every job's samples delivered to the caller's planar host arrays (the host link is the bound)"}
            del hbuf
        full_chip = None
        if world == 1 and args.full_chip and inst < 256 :
            # the half-empty chip is the configuration's choice (1024 jobs over 8 GPUs = 128 per GPU, one island per CU): the same jobs at one per CU
            try:
                rt2 = Runtime(graphs.C4_SAMPLE_RATE, BLOCK, device=local)
                rt2.set_option("batch_blocks", B)
                rt2.set_option("specialize", args.specialize)
                assert rt2.render(*[graphs.c4_instance(k) for k in range(256)])["result"] == 0
                out2 = torch.zeros((B, 256, BLOCK), dtype=torch.float32, device="cuda")
                for _ in range(2):
                    rt2.process_blocks(B, 256, out_ptr=out2.data_ptr())
                torch.cuda.synchronize()
                fs = max(2, min(args.steps, 6))
                t1 = time.perf_counter()
                for _ in range(fs):
                    rt2.process_blocks(B, 256, out_ptr=out2.data_ptr())
                torch.cuda.synchronize()
                dtf = time.perf_counter() - t1
                full_chip = {"instances": 256, "value": 256 * BLOCK * fs * B / dtf, "unit": "samples/s", "ms_per_step": 1e3 * dtf / fs, "steps": fs,
                             "mode": "device-resident; 256 jobs = one island per CU of the whole chip (not a BASELINE geometry; not parity-checked here: tests/test_gpu_baseline_sizes.py renders the same jobs)"}
                del rt2, out2
            except Exception as e:      # noqa: BLE001
                full_chip = {"error": repr(e)[:200]}
        base = None if args.no_cpu_baseline else c4_cpu_baseline(inst, inst * rank)
        traffic, traffic_src = None, None
        tpath = os.path.join(ROOT, "profiles", "traffic_c4.json")
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                if int(tj.get("instances", 0)) == inst and int(tj.get("blocks_per_launch", 0)) == B:
                    traffic = tj.get("hbm_bytes_per_launch_set")
                    traffic_src = "profiles/traffic_c4.json (rocprofv3 PMC passes of this command, committed; not re-measured in this run): " + str(tj.get("round", ""))
            except Exception:
                traffic = None
        alg = graphs.c4_algorithmic_bytes(inst)                  # per block-step of one rank
        us = 1e6 * dt / blocks
        sets = max(1, prof["launch_sets"])
        stats = rt.stats()
        _emit({
            "metric": "instance-samples/sec, independent offline render instances (BASELINE configs[3])",
            "value": world * inst * BLOCK * blocks / dt, "unit": "samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": f"BASELINE configs[3] (C4): {inst} independent render instances per GPU ({inst * world} in all), "
                                   "each rand -> svf -> delay{24000} -> biquad -> sdelay -> tanh (odd instances: biquad before svf), "
                                   f"sr 48000, blockSize 512; one step = one launch set of {B} blocks of every instance",
                       "instances_per_gpu": inst, "instances_total": inst * world, "ranks_seen": ranks_seen,
                       "blocks_per_step": B, "islands": stats["num_islands"], "launch_levels": stats["num_levels"],
                       "mode": "host buffers (elemhip_process_blocks_host): every job's samples delivered to the caller's arrays" if host_mode
                               else "device-resident render (elemhip_process_blocks): the jobs' samples stay in HBM; `host_delivered` = the same through elemhip_process_blocks_host",
                       "collectives": ("none on one GPU" if world == 1 else
                                       "RCCL gather of every chunk's per-job outputs on rank 0 inside the timed region "
                                       f"({gathered_elems * 4 / max(1, args.steps) / 1e6:.1f} MB per step); nothing is exchanged while rendering")},
            "us_per_block_step": us, "plan_build_ms": build_ms, "host_delivered": host_delivered, "full_chip_256_jobs": full_chip,
            "instances_256_samples_per_s": (full_chip or {}).get("value"), "host_delivered_samples_per_s": (host_delivered or {}).get("value"),
            "roofline": {"bound": "hbm", "achieved": alg / (us * 1e-6) / 1e9, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": alg / (us * 1e-6) / 1e9 / HBM_PEAK_GBPS, "traffic": traffic, "traffic_source": traffic_src,
                         "algorithmic_bytes_per_block_step": alg, "algorithmic_bytes_per_launch_set": alg * B,
                         "launch_us_per_step": [1e3 * x / sets for x in prof["level_ms"]],
                         "note": "bound by the biquad recurrence wave of a job island (one lane of one wave: 8 VALU slots per frame, DESIGN 4), on 128 of "
                                 "256 CUs; `traffic` = PMC HBM bytes of one launch set (2 x FETCH_SIZE + WRITE_SIZE, MI355X_MICROARCH.md)"},
            "cpu_baseline": base, "speedup_vs_cpu_baseline": (world * inst * BLOCK * blocks / dt) / base["value"] if base else None,
            "speedup_vs_cpu_baseline_note": (f"GPU rate / the reference engine on ALL {base['cores']} host cores this process may use" if base else None),
            "parity": parity,
        })
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
    ap.add_argument("--no-configs", action="store_true", help="skip the sub-records of the other configurations (C1, C3, C4, C5, C5 shape churn, taps)")
    ap.add_argument("--configs", default="c1,c3,c4,c5,c5_churn,taps", help="which sub-records to append (comma separated)")
    ap.add_argument("--config-timeout", type=float, default=240.0, help="seconds a sub-record's process may take before it is given up")
    ap.add_argument("--backend", choices=["nccl", "gloo"], default="nccl",
                    help="process-group backend for N > 1 ('nccl' = RCCL; 'gloo' lets a test run two ranks on ONE GPU)")
    ap.add_argument("--shared-gpu", action="store_true", help="testing: every rank uses cuda:0 (needs --backend gloo)")
    ap.add_argument("--check", action="store_true",
                    help="N > 1, c2: compare the last reduced chunk on rank 0 with the reference engine rendering the whole graph")
    ap.add_argument("--device-resident", action="store_true",
                    help="time elemhip_process_blocks with the output bus left in HBM (the r01/r02 protocol) instead of the host-buffer entry")
    ap.add_argument("--host-delivered", action="store_true", help="--workload c4: time elemhip_process_blocks_host (r03-r05's protocol) instead of the device-resident render")
    ap.add_argument("--no-host-leg", action="store_true", help="--workload c4: skip the host-delivered sub-record")
    ap.add_argument("--no-full-chip", dest="full_chip", action="store_false", help="--workload c4: skip the 256-job sub-record")
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
    if world != args.gpus and (args.gpus > 1 or world > 1):
        raise SystemExit(f"--gpus {args.gpus} needs WORLD_SIZE={args.gpus} (launch with torch.distributed.run); got {world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP engine has no CPU fallback")
    if args.shared_gpu:
        local = 0
    torch.cuda.set_device(local)
    ranks_seen = 1
    if world > 1:
        dist.init_process_group(args.backend, rank=rank, world_size=world)   # "nccl" == RCCL on ROCm
        ranks_seen = dist.get_world_size()
        if ranks_seen != args.gpus:       # the driver computes scaling from n_gpus: never print a line for another job size
            raise SystemExit(f"--gpus {args.gpus} but the process group has {ranks_seen} ranks")

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

    import numpy as np
    spc = max(1, args.steps_per_call)
    host_mode = world == 1 and not args.device_resident     # the headline mode: samples land in host memory
    bufs = [torch.zeros((spc * B, 2, BLOCK), dtype=torch.float32, device="cuda") for _ in range(2)]
    pinned = torch.zeros((spc * B, 2, BLOCK), dtype=torch.float32).pin_memory()

    last_chunk = {}

    def run_device(steps: int) -> None:
        """elemhip_process_blocks: outputs stay in HBM (N > 1: async RCCL sum-reduce of every chunk to rank 0, which copies
        the reduced bus to pinned host memory)."""
        done, k, works = 0, 0, []
        while done < steps:
            c = min(spc, steps - done)
            buf = bufs[k % 2]
            if world > 1 and len(works) >= 2:
                w, b_ = works.pop(0)
                w.wait()                       # the buffer we are about to overwrite has been reduced
                if rank == 0:
                    pinned[:b_.shape[0]].copy_(b_, non_blocking=True)
                torch.cuda.current_stream().synchronize()   # the engine renders on its own stream
            rt.process_blocks(c * B, 2, out_ptr=buf.data_ptr())
            if world > 1:
                works.append((reduce_bus(buf[:c * B], dst=0, async_op=True), buf[:c * B]))
            else:           # the same delivery at N = 1: the chunk goes to rank 0's pinned host memory (no reduce to wait for)
                pinned[:c * B].copy_(buf[:c * B], non_blocking=True)
            done += c
            k += 1
        for w, b_ in works:
            w.wait()
            if rank == 0:
                pinned[:b_.shape[0]].copy_(b_, non_blocking=True)
        if world > 1 and rank == 0 and works:
            torch.cuda.synchronize()
            last_chunk["blocks"] = int(works[-1][1].shape[0])

    def run_host(steps: int, out) -> None:
        """elemhip_process_blocks_host: ONE call renders `steps` launch sets into the planar host array out[2, frames]."""
        rt.process_blocks_host(None, 2, steps * B * BLOCK, out=out)

    warm_host = np.empty((2, max(1, args.warmup) * B * BLOCK), dtype=np.float32) if host_mode else None
    timed_host = np.empty((2, args.steps * B * BLOCK), dtype=np.float32) if host_mode else None
    if host_mode:
        timed_host.fill(0.0)                                     # touch the pages outside the timed region
        if args.warmup:
            run_host(args.warmup, warm_host)
    else:
        run_device(args.warmup)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    rt.set_option("profile_launches", 1)            # HIP event pair around every launch of the timed region
    t0 = time.perf_counter()
    if host_mode:
        run_host(args.steps, timed_host)
    else:
        run_device(args.steps)
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
        traffic, traffic_src = None, None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                traffic = tj.get("c2_hbm_bytes_per_block")
                traffic = traffic * B if traffic else None            # per launch set, like `achieved`'s basis
                traffic_src = "profiles/traffic.json (rocprofv3 PMC passes of this command, committed; not re-measured in this run): " + str(tj.get("round", ""))
            except Exception:
                traffic = None
        # ---- the device-resident rate (no delivery), same engine, right after the timed region ----
        device_resident = None
        if host_mode:
            run_device(min(args.warmup, 2))
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            run_device(args.steps)
            torch.cuda.synchronize()
            dtd = time.perf_counter() - t1
            device_resident = {"value": graph_frames / dtd, "unit": "samples/s", "ms_per_step": 1e3 * dtd / args.steps,
                               "us_per_block": 1e6 * dtd / blocks,
                               "mode": "THE N > 1 PROTOCOL AT N = 1: elemhip_process_blocks per step, device-resident render, every step's bus chunk copied "
                                       "asynchronously to rank 0's pinned host memory (N > 1 adds the RCCL sum-reduce of the chunk to rank 0 in front of that copy). "
                                       "A 1 -> N curve is self-consistent when its N = 1 point is THIS figure; the headline `value` at N = 1 is the stricter "
                                       "delivery into the caller's planar host arrays (elemhip_process_blocks_host)"}
        # ---- latency figures outside the timed region ----
        rt.set_option("time_batch", 1)
        lv1 = rt.time_launches(2, 100)
        for _ in range(20):
            rt.process(None, 2, BLOCK)
        t1 = time.perf_counter()
        for _ in range(200):
            rt.process(None, 2, BLOCK)
        sync_us = 1e6 * (time.perf_counter() - t1) / 200
        # the same call from a native host (examples/bench_cli: the reference's cli/Benchmark.cpp protocol, C++ over the C-ABI, no
        # Python between the calls), rank 0 only, outside the timed region
        sync_native = None
        if rank == 0:
            try:
                sys.path.insert(0, os.path.join(ROOT, "benchmarks"))
                import bench_configs as _bc
                sync_native = _bc._native_host(graphs.c2_graph(voices=my_voices, channels=2, first_voice=first), graphs.C2_SAMPLE_RATE, blocks=2000,
                                               env={"ELEMHIP_SPECIALIZE": "2"})
                # ... and at the PRODUCT default (background compilation: the kernel cache on disk is warm here, so the specialised kernels
                # are there from the first blocks; a cold cache renders through the interpreter kernels until the compiles finish)
                if isinstance(sync_native, dict) and "error" not in sync_native:
                    sync_native["specialize"] = 2
                    sync_native["product_default_specialize_1"] = _bc._native_host(
                        graphs.c2_graph(voices=my_voices, channels=2, first_voice=first), graphs.C2_SAMPLE_RATE, blocks=2000, env={"ELEMHIP_SPECIALIZE": "1"})
                    # ... and opt-in `resident`: the calls handed to a kernel that stays on the GPU (interpreter island bodies behind
                    # device-wide barriers between this graph's levels; resident.hip)
                    sync_native["resident_opt_in"] = _bc._native_host(
                        graphs.c2_graph(voices=my_voices, channels=2, first_voice=first), graphs.C2_SAMPLE_RATE, blocks=2000,
                        env={"ELEMHIP_SPECIALIZE": "1", "ELEMHIP_RESIDENT": "1"})
            except Exception as e:      # noqa: BLE001  (a missing binary is reported, not fatal)
                sync_native = {"error": str(e)[:200]}

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
                "workload": ("BASELINE configs[1] (C2): 256-voice subtractive synth, 4107 nodes " if my_voices == 256 else
                             f"C2 voice graph at {my_voices} voices ({my_voices * 16 + 11} nodes; BASELINE configs[1] is the 256-voice case) ")
                            + "(2 blepsaw, train gate, pole envelope, svf lowpass, tanh per voice; 2 mix adds, 2 roots), "
                            "sr 48000, blockSize 512, 0 in / 2 out"
                            + (", per GPU" if args.scaling == "weak" and world > 1 else ""),
                "step": f"one launch set = {B} consecutive 512-frame blocks of the whole graph ({B * BLOCK} output frames)",
                "blocks_per_step": B,
                "frames_per_step": B * BLOCK,
                "nodes_per_gpu": stats["num_nodes_in_plan"],
                "voices_per_gpu": my_voices,
                "voices_total": total_voices, "ranks_seen": ranks_seen,
                "block_size": BLOCK,
                "mode": ("host buffers: elemhip_process_blocks_host, every block delivered to the caller's planar host arrays "
                         "(pinned double-buffered launch sets, D2H of set k under the rendering of set k + 1); all K steps in one call")
                        if host_mode else
                        ("offline elemhip_process_blocks, output bus resident in HBM"
                         + (", RCCL sum-reduce of the bus to rank 0 per call, rank 0 copies the reduced bus to pinned host memory" if world > 1 else "")),
                "steps_per_call": args.steps if host_mode else spc,
                "pipelined_blocks_in_flight": rt.describe_plan()["islands"][0]["copies"],
                "voices_per_island": rt.describe_plan()["pack_k"],
                "islands": stats["num_islands"], "launch_levels": stats["num_levels"], "max_lds_bytes": stats["max_lds_bytes"],
                "island_kernels": ("run-time specialised per island shape (hiprtc, gfx950): %d shape(s) covering %d islands, %d launches; "
                                   "compile wait %.0f ms inside plan_build_ms (0 = on-disk cache hit)"
                                   % (stats["spec_shapes"], stats["spec_islands"], stats["spec_launches"], stats["last_jit_wait_ms"]))
                                  if args.specialize and stats["spec_launches"] else "ahead-of-time interpreter kernel",
                "multi_gpu_note": "no 1 -> 8 GPU curve has been measured by the builder (single-GPU boxes only); N > 1 is covered by 2- and 8-process tests "
                                  "(gloo on CPU, engines sharing one GPU). N > 1 lines deliver through device-resident render + RCCL reduce + copy to rank 0's "
                                  "pinned memory; the N = 1 line's `device_resident` sub-record is that same protocol at N = 1",
            },
            "us_per_block": us_per_block,
            "realtime_factor_48k": value / 48000.0,
            "plan_build_ms": build_ms,
            "sync_process_us_per_block": sync_us,
            "sync_process_native_host": sync_native,
            "device_resident": device_resident,
            "roofline": {
                "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic, "traffic_source": traffic_src,
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
            head_blocks = min(B, (args.warmup if args.warmup else args.steps) * B) if host_mode else 0
            cb = cpu_baseline(keep_blocks=head_blocks, voices=my_voices)
            if cb:
                head = cb.pop("head")
                out["cpu_baseline"] = cb
                out["speedup_vs_cpu_baseline"] = out["value"] / cb["value"]
                parity = {"tolerance": 1e-6, "unit": "abs (x max|ref| when > 1)", "checker": cb["kind"]}
                if host_mode and head is not None:
                    got = (warm_host if args.warmup else timed_host)[:, :head.shape[1]]
                    scale = max(1.0, float(np.abs(head).max()))
                    parity["head"] = {"blocks": head.shape[1] // BLOCK, "max_abs_err": float(np.abs(got - head).max()), "scale": scale,
                                      "what": "blocks 0.. of the first launch set (%s) vs the reference engine on one core" % ("warm-up" if args.warmup else "timed")}
                try:
                    mc = cpu_baseline_multicore(cb["ms_per_block"], advance_blocks=(args.warmup + args.steps) * B if host_mode else 0, tail=64, voices=my_voices)
                except Exception as e:   # the fairness variant must never cost the headline line
                    mc = {"error": repr(e)}
                if mc:
                    tail = mc.pop("tail", None)
                    out["cpu_baseline_all_cores"] = mc
                    if "value" in mc:
                        out["speedup_vs_cpu_all_cores"] = out["value"] / mc["value"]
                    if host_mode and tail is not None:
                        got = timed_host[:, -tail.shape[1]:].astype(np.float64)
                        scale = max(1.0, float(np.abs(tail).max()))
                        parity["tail"] = {"blocks": tail.shape[1] // BLOCK, "max_abs_err": float(np.abs(got - tail).max()), "scale": scale,
                                          "what": "the LAST blocks of the LAST timed step vs the reference engine advanced through all "
                                                  "%d blocks (voices partitioned over %d host processes, partial buses summed in voice order in float64)"
                                                  % ((args.warmup + args.steps) * B, mc.get("cores", 0))}
                    elif host_mode:
                        parity["tail"] = None
                        parity["tail_skipped"] = "advancing the reference through the whole run would not fit the time budget on this host"
                errs = [parity[k]["max_abs_err"] / parity[k]["scale"] for k in ("head", "tail") if parity.get(k)]
                out["parity_max_abs_err"] = max(parity[k]["max_abs_err"] for k in ("head", "tail") if parity.get(k)) if errs else None
                parity["ok"] = bool(errs) and max(errs) <= 1e-6
                out["parity"] = parity
        if world > 1 and args.check and last_chunk:
            # the whole graph (every rank's voices) on the reference engine, advanced to the last reduced chunk
            import oracle
            chk = oracle.RefRuntime(graphs.C2_SAMPLE_RATE, BLOCK) if oracle.have_ref() else oracle.PortRuntime(graphs.C2_SAMPLE_RATE, BLOCK)
            assert chk.render(*graphs.c2_graph(voices=total_voices, channels=2, first_voice=0))["result"] == 0
            nb = last_chunk["blocks"]
            total_blocks = (args.warmup + args.steps) * B
            for _ in range(total_blocks - nb):
                chk.process(None, 2, BLOCK)
            ref = np.stack([chk.process(None, 2, BLOCK) for _ in range(nb)])
            got = pinned[:nb].numpy()
            scale = max(1.0, float(np.abs(ref).max()))
            err = float(np.abs(got - ref).max())
            out["parity_max_abs_err"] = err
            out["parity"] = {"tolerance": 1e-6, "ok": err <= 1e-6 * scale, "scale": scale,
                             "what": f"last {nb} blocks of the rank-0 reduced bus vs the reference engine rendering all {total_voices} voices"}
        if world == 1 and not args.no_configs and my_voices == 256 and not args.device_resident:
            out["configs"] = run_configs([c for c in args.configs.split(",") if c], args.config_timeout)
        _emit(out)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
