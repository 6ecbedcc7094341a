"""N>1 path on CPU: two processes over gloo, each rendering its shard of the voices with a CPU
checker standing in for the per-GPU engine; the bus exchange must reproduce the single-engine mix."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import oracle

pytestmark = pytest.mark.skipif(not (oracle.have_port() or oracle.have_ref()), reason="needs a CPU checker")

VOICES, BLOCKS, BS = 12, 6, 512


def _engine(sr, bs):
    return oracle.PortRuntime(sr, bs) if oracle.have_port() else oracle.RefRuntime(sr, bs)


def _render(first, count):
    from elementary_amd import graphs
    rt = _engine(48000.0, BS)
    assert rt.render(*graphs.c2_graph(voices=count, channels=2, first_voice=first))["result"] == 0
    return np.stack([rt.process(None, 2, BS) for _ in range(BLOCKS)])


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from elementary_amd.sharded import gather_outputs, ordered_bus_sum, reduce_bus, shard_range
    b, e = shard_range(VOICES, world, rank)
    bus = torch.from_numpy(_render(b, e - b))
    ordered = ordered_bus_sum(bus)
    per_unit = gather_outputs(torch.full((e - b, 3), float(rank)))
    reduce_bus(bus)
    if rank == 0:
        q.put((bus.numpy(), ordered.numpy(), per_unit.numpy()))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_bus_exchange_matches_single_engine():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    reduced, ordered, per_unit = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    from elementary_amd.sharded import shard_range
    parts = [_render(*(lambda b, e: (b, e - b))(*shard_range(VOICES, 2, r))) for r in range(2)]
    assert np.array_equal(ordered, parts[0] + parts[1])           # rank-ordered sum is reproducible
    assert np.allclose(reduced, parts[0] + parts[1], atol=1e-6)
    # voices alternate channels by index parity, so a shard boundary at an even voice keeps the
    # per-channel voice sets of the sharded and unsharded synth identical: same signal, other fold order
    full = _render(0, VOICES)
    assert np.abs(full - ordered).max() <= 1e-6
    assert per_unit.shape == (VOICES, 3) and per_unit[:6].max() == 0 and per_unit[6:].min() == 1


def test_shard_range_partitions_exactly():
    from elementary_amd.sharded import shard_range
    for n in (0, 1, 7, 256, 1024):
        for w in (1, 2, 3, 8):
            r = [shard_range(n, w, k) for k in range(w)]
            assert r[0][0] == 0 and r[-1][1] == n
            assert all(r[k][1] == r[k + 1][0] for k in range(w - 1))
            sizes = [e - b for b, e in r]
            assert max(sizes) - min(sizes) <= 1


# ---- BASELINE configs[3]: independent offline-render instances sharded over the ranks, outputs gathered ----------
INSTANCES, C4_BLOCKS = 6, 5


def _render_instances(first, count):
    """Each rank renders its instances through the offline-render caller (offline-renderer/index.ts:87-133)."""
    from elementary_amd import graphs
    from elementary_amd.offline import OfflineRenderer
    r = OfflineRenderer(_engine)
    r.initialize(num_input_channels=0, num_output_channels=count, sample_rate=graphs.C4_SAMPLE_RATE, block_size=BS)
    r.render(*[graphs.c4_instance(k) for k in range(first, first + count)])
    outs = [np.zeros(C4_BLOCKS * BS, dtype=np.float32) for _ in range(count)]
    r.process([], outs)
    return np.stack(outs)                                       # [instances, frames]


def _c4_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from elementary_amd.sharded import gather_outputs, shard_range
    b, e = shard_range(INSTANCES, world, rank)
    gathered = gather_outputs(torch.from_numpy(_render_instances(b, e - b)))
    if rank == 0:
        q.put(gathered.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_instance_gather_matches_single_engine():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_c4_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    gathered = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    full = _render_instances(0, INSTANCES)
    assert gathered.shape == full.shape
    assert np.array_equal(gathered, full)        # instances are independent: sharding changes nothing, bit for bit
