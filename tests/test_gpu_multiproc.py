"""The HIP engine under a process group, on the hardware the driver has: TWO ranks sharing ONE GPU (gloo backend, device
tensors), running bench.py's own N > 1 code path — per-rank engines on their own streams, double-buffered device buses,
asynchronous bus reduce with buffer reuse, the C4 output gather — and checked against the reference engine rendering
the whole graph. (RCCL itself refuses two ranks on one device; the 8-GPU run is the driver's.)"""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _launch(extra, ranks=2):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", ELEMHIP_SPECIALIZE="2")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(ranks), "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", str(ranks), "--backend", "gloo", "--shared-gpu",
           "--no-cpu-baseline"] + extra
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert res.returncode == 0, (res.stdout[-2000:], res.stderr[-3000:])
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    # rank 0 prints its full record on a line of its own, then ONE compact line (< 4 KB) LAST: the line the driver parses
    assert len(lines) == 2 and json.loads(lines[0]).get("record") == "headline_full", res.stdout[-2000:]
    assert len(lines[-1].encode()) < 4096 and res.stdout.rstrip().endswith(lines[-1])
    return json.loads(lines[-1])


@pytest.mark.parametrize("scaling", ["weak", "strong"])
def test_two_ranks_one_gpu_bus_reduce(gpu_required, scaling):
    out = _launch(["--steps", "3", "--warmup", "1", "--batch-blocks", "64", "--voices", "24", "--scaling", scaling,
                   "--steps-per-call", "1", "--check"])
    assert out["n_gpus"] == 2 and out["scaling"] == scaling
    assert out["config"]["voices_total"] == (48 if scaling == "weak" else 24)
    assert out["parity"]["ok"], out["parity"]
    assert out["value"] > 0


def test_two_ranks_one_gpu_c4_gather(gpu_required):
    out = _launch(["--workload", "c4", "--instances", "6", "--steps", "2", "--warmup", "1", "--batch-blocks", "32"])
    assert out["n_gpus"] == 2 and out["config"]["instances_total"] == 12
    assert "gather" in out["config"]["collectives"]
    assert out["value"] > 0


def test_eight_ranks_one_gpu_both_workloads(gpu_required):
    """The rank count the driver launches (8), on the hardware there is (one GPU, gloo): eight engines with their own streams,
    pinned buffers and kernel caches side by side, the bus reduce / the output gather across eight ranks, the line's
    `ranks_seen` = what the process group reports. No 1 -> 8 GPU curve exists: this checks the code path, not scaling."""
    out = _launch(["--steps", "2", "--warmup", "1", "--batch-blocks", "32", "--voices", "8", "--steps-per-call", "1", "--check"], ranks=8)
    assert out["n_gpus"] == 8 and out["config"]["ranks_seen"] == 8 and out["config"]["voices_total"] == 64
    assert out["parity"]["ok"], out["parity"]
    out = _launch(["--workload", "c4", "--instances", "3", "--steps", "2", "--warmup", "1", "--batch-blocks", "16"], ranks=8)
    assert out["n_gpus"] == 8 and out["config"]["ranks_seen"] == 8 and out["config"]["instances_total"] == 24


def test_bench_refuses_a_world_that_is_not_gpus():
    """`--gpus N` with a process group of another size prints no line (the driver computes scaling from n_gpus)."""
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "1", "--warmup", "0"], capture_output=True, text=True,
                         timeout=120, cwd=ROOT, env=env)
    assert res.returncode != 0 and not [ln for ln in res.stdout.splitlines() if ln.startswith("{")]


def test_sum_buses_is_the_rank_ordered_sum(gpu_required):
    """elemhip_sum_buses (the C-ABI's multi-GPU helper): the partial buses of voice shards rendered by separate engines, added in
    rank order on the device — the bits of a sequential float32 sum in that order (and of sharded.ordered_bus_sum), for 2, 5 and 64
    partials, a length that is not a multiple of the workgroup size, and through the error paths."""
    import torch
    from elementary_amd import graphs
    from elementary_amd.runtime import Runtime, sum_buses, ElemHipError
    shards = []
    for r in range(5):                      # five "ranks": 8 voices each of a 40-voice graph
        rt = Runtime(graphs.C2_SAMPLE_RATE, 512, device=0)
        assert rt.render(*graphs.c2_graph(voices=8, channels=2, first_voice=8 * r))["result"] == 0
        out = torch.zeros((16, 2, 512), dtype=torch.float32, device="cuda")
        rt.process_blocks(16, 2, out_ptr=out.data_ptr())
        shards.append(out)
    torch.cuda.synchronize()
    n = shards[0].numel()
    dst = torch.empty_like(shards[0])
    for count in (2, 5):
        sum_buses(dst.data_ptr(), [s.data_ptr() for s in shards[:count]], n)
        torch.cuda.synchronize()
        ref = shards[0].cpu().numpy().copy()
        for s in shards[1:count]:
            ref = ref + s.cpu().numpy()                       # float32, rank order
        assert np.array_equal(dst.cpu().numpy(), ref)
    many = [torch.randn(1000 + 37, device="cuda") for _ in range(64)]
    d2 = torch.empty(1037, device="cuda")
    sum_buses(d2.data_ptr(), [m.data_ptr() for m in many], 1037)
    torch.cuda.synchronize()
    ref = many[0].cpu().numpy().copy()
    for m in many[1:]:
        ref = ref + m.cpu().numpy()
    assert np.array_equal(d2.cpu().numpy(), ref)
    with pytest.raises(ElemHipError):
        sum_buses(d2.data_ptr(), [m.data_ptr() for m in many] + [many[0].data_ptr()], 1037)      # 65 partials
    with pytest.raises(ElemHipError):
        sum_buses(d2.data_ptr(), [], 1037)
