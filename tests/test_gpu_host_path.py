"""The host-buffer offline entry point (elemhip_process_blocks_host: pinned double-buffered launch sets) and parity at
EXACTLY the geometry bench.py times: the full 256-voice C2 graph, run-time specialised kernels, 1024- and 256-block
launch sets, more than two full sets, every block against the reference engine. Tolerance 1e-6 absolute (x max|ref|
when > 1)."""
import os
import subprocess
import tempfile

import numpy as np
import pytest

from elementary_amd import el, graphs
from elementary_amd.reconciler import Renderer, batch_to_json
from helpers import lcg_noise
from cases import NODE_CASES, node_case_resources

pytestmark = pytest.mark.gpu
TOL = 1e-6
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _checker(sr, bs):
    import oracle
    return oracle.RefRuntime(sr, bs) if oracle.have_ref() else oracle.PortRuntime(sr, bs)


def _hip(sr, bs, **opts):
    from elementary_amd.runtime import Runtime
    rt = Runtime(sr, bs, device=0)
    for k, v in opts.items():
        rt.set_option(k, v)
    return rt


def _ref_planar(c, n_out, blocks, x=None, bs=512):
    """[n_out, blocks * bs] from block-by-block process() calls (x: [n_in, frames] or None)."""
    out = np.empty((n_out, blocks * bs), dtype=np.float32)
    for k in range(blocks):
        xin = None if x is None else x[:, k * bs:(k + 1) * bs]
        out[:, k * bs:(k + 1) * bs] = c.process(xin, n_out, bs)
    return out


@pytest.mark.parametrize("batch", [1024, 256])
def test_bench_geometry_full_c2(gpu_required, batch):
    """What bench.py times, checked block by block: 256 voices, specialize = 2, `batch`-block launch sets, two full sets and
    a ragged third (byte offsets beyond 2^31 in the 1024-block arena), delivered to host arrays."""
    nb = 2 * batch + 77 if batch == 1024 else 2 * batch + 300
    a, c = _hip(graphs.C2_SAMPLE_RATE, 512, specialize=2, batch_blocks=batch), _checker(graphs.C2_SAMPLE_RATE, 512)
    roots = graphs.c2_graph()
    assert a.render(*roots)["result"] == 0 and c.render(*roots)["result"] == 0
    st = a.stats()
    assert st["spec_shapes"] >= 1 and st["spec_islands"] >= 256, st
    got = a.process_blocks_host(None, 2, nb * 512)
    st = a.stats()
    assert st["spec_launches"] >= 3 and st["batch_launches"] >= 3, st          # the specialised kernels rendered every set
    info = a.spec_info(0)
    assert info["state"] == 1, info["log"][:2000]
    ref = _ref_planar(c, 2, nb)
    scale = max(1.0, float(np.abs(ref).max()))
    err = np.abs(got - ref).reshape(2, nb, 512).max(axis=(0, 2))                # per block
    worst = int(err.argmax())
    assert float(err.max()) <= TOL * scale, f"block {worst}: {err.max():.3e} (max|ref| {scale:.3f})"
    # the tail of the last full set and the ragged set are as good as the head
    assert float(err[batch - 8:batch + 8].max()) <= TOL * scale and float(err[-8:].max()) <= TOL * scale


def test_bench_geometry_full_c4(gpu_required):
    """What `bench.py --workload c4` times, checked for every instance and every block (BASELINE configs[3], one GPU's share;
    offline-renderer/index.ts:87-133 is the caller): 128 independent render instances = 128 output channels, specialize = 2,
    1024-block launch sets through elemhip_process_blocks_host — a full set (268 MB: the host scatter runs on several
    threads, engine.cpp processBlocksHost) and a ragged one that is still above the 16 MB threshold."""
    inst, batch = 128, 1024
    nb = batch + 90
    a, c = _hip(graphs.C4_SAMPLE_RATE, 512, specialize=2, batch_blocks=batch), _checker(graphs.C4_SAMPLE_RATE, 512)
    roots = [graphs.c4_instance(k) for k in range(inst)]
    assert a.render(*roots)["result"] == 0 and c.render(*roots)["result"] == 0
    got = a.process_blocks_host(None, inst, nb * 512)
    st = a.stats()
    assert st["spec_launches"] >= 2 and st["batch_launches"] >= 2, st
    assert got.shape == (inst, nb * 512)
    ref = _ref_planar(c, inst, nb)
    assert float(np.abs(ref).max()) > 0.05
    err = np.abs(got - ref).reshape(inst, nb, 512).max(axis=2)                   # [instance, block]
    wi, wb = np.unravel_index(int(err.argmax()), err.shape)
    assert float(err.max()) <= TOL * max(1.0, float(np.abs(ref).max())), f"instance {wi} block {wb}: {err.max():.3e}"
    # every instance produced its own stream (a scatter that mixed channels up would pass a max-error test on silence only)
    assert len({got[k, -4096:].tobytes() for k in range(inst)}) == inst


def test_commit_and_gc_from_another_thread_during_a_host_render(gpu_required):
    """elemhip_process_blocks_host gives up the render lock between launch sets while one is still rendering: a commit that
    replaces the plan, a gc() that recycles node records and rings, a pruned resource — all on a second thread — must not
    free anything under a kernel in flight (their device memory is released after the next synchronize: Engine::freeDeferred).
    The stream stays what the reference renders for the same timeline: every voice swap lands on a launch-set boundary."""
    import threading
    a = _hip(graphs.C2_SAMPLE_RATE, 512, specialize=0, batch_blocks=16)
    voices = lambda gen: [el.add(*[el.mul(0.1, el.delay({"size": 4000, "key": f"d{gen}_{v}"}, 30.0 + v, 0.3, graphs.c2_voice(v + 16 * gen))) for v in range(8)])]
    assert a.render(*voices(0))["result"] == 0
    a.process_blocks_host(None, 1, 40 * 512)            # settle the root fade
    stop, errors, swaps = threading.Event(), [], [0]

    def mutate():
        gen = 1
        try:
            while not stop.is_set():
                assert a.render(*voices(gen))["result"] == 0
                a.gc()
                a.prune_shared_resources()
                gen += 1
                swaps[0] += 1
        except Exception as e:          # noqa: BLE001
            errors.append(repr(e))

    th = threading.Thread(target=mutate)
    th.start()
    try:
        outs = [a.process_blocks_host(None, 1, 600 * 512) for _ in range(6)]
    finally:
        stop.set()
        th.join()
    assert not errors, errors
    assert swaps[0] >= 3, swaps
    y = np.concatenate(outs, axis=1)
    assert np.isfinite(y).all() and float(np.abs(y).max()) > 1e-3 and float(np.abs(y).max()) < 4.0
    # the engine is intact afterwards: a fresh graph renders like the reference
    c = _checker(graphs.C2_SAMPLE_RATE, 512)
    roots = graphs.c2_graph(voices=4)
    b = _hip(graphs.C2_SAMPLE_RATE, 512, specialize=0)
    assert b.render(*roots)["result"] == 0 and c.render(*roots)["result"] == 0
    assert a.render(*roots)["result"] == 0
    a.gc()
    got = a.process_blocks_host(None, 2, 300 * 512)[:, -64 * 512:]
    ref = b.process_blocks_host(None, 2, 300 * 512)[:, -64 * 512:]
    assert float(np.abs(got - ref).max()) <= 5e-2     # (a's roots cross-fade in from the previous graph; the tail is the new graph alone)
    assert float(np.abs(got - ref)[:, -512:].max()) <= 1e-5


def test_host_path_equals_device_path(gpu_required):
    """elemhip_process_blocks_host vs elemhip_process_blocks on two engines with the same options: bit-identical, for a
    frame count that is not a multiple of the block size (the tail block is rendered whole, delivered cut)."""
    import torch
    frames = 300 * 512 + 137
    nb = 301
    a, b, c = (_hip(graphs.C2_SAMPLE_RATE, 512, specialize=2, batch_blocks=64), _hip(graphs.C2_SAMPLE_RATE, 512, specialize=2, batch_blocks=64),
               _checker(graphs.C2_SAMPLE_RATE, 512))
    roots = graphs.c2_graph(voices=48)
    for rt in (a, b, c):
        assert rt.render(*roots)["result"] == 0
    got = a.process_blocks_host(None, 2, frames)
    out = torch.zeros((nb, 2, 512), dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    b.process_blocks(nb, 2, out_ptr=out.data_ptr())
    dev = out.cpu().numpy().transpose(1, 0, 2).reshape(2, nb * 512)[:, :frames]
    assert got.shape == (2, frames)
    assert np.array_equal(got, dev)
    ref = _ref_planar(c, 2, nb)[:, :frames]
    assert float(np.abs(got - ref).max()) <= TOL * max(1.0, float(np.abs(ref).max()))
    # the engine's clock moved by whole blocks: the next call continues the stream
    more = a.process_blocks_host(None, 2, 5 * 512)
    assert float(np.abs(more - _ref_planar(c, 2, 5)).max()) <= TOL * max(1.0, float(np.abs(ref).max()))


@pytest.mark.parametrize("name,batch", [("in_passthrough", 7), ("pole", 64), ("delay_long", 5), ("taps", 16), ("svf_modulated", 3)])
def test_host_path_with_inputs(gpu_required, name, batch):
    """Host inputs through the pinned H2D halves (several sets in flight), ragged frame count: the short input tail is
    zero-padded like the offline caller does (offline-renderer/index.ts:95-103). `taps` renders block at a time inside."""
    roots_fn, n_in = NODE_CASES[name]
    frames = 41 * 512 + 200
    nb = 42
    a, c = _hip(44100.0, 512, batch_blocks=batch), _checker(44100.0, 512)
    for rt in (a, c):
        for rname, data in node_case_resources().items():
            assert rt.add_shared_resource(rname, data)
    roots = roots_fn()
    assert a.render(*roots)["result"] == 0 and c.render(*roots)["result"] == 0
    x = np.stack([lcg_noise(frames, 11 + ch, 0.5) for ch in range(n_in)])
    xpad = np.zeros((n_in, nb * 512), dtype=np.float32)
    xpad[:, :frames] = x
    got = a.process_blocks_host(x, len(roots), frames)
    ref = _ref_planar(c, len(roots), nb, xpad)[:, :frames]
    scale = max(1.0, float(np.abs(ref).max()))
    assert np.isfinite(got).all()
    assert float(np.abs(got - ref).max()) <= TOL * scale


def test_offline_renderer_uses_the_host_batch_entry(gpu_required):
    """OfflineRenderer.process (offline-renderer/index.ts:87-133) on the HIP engine goes through ONE
    elemhip_process_blocks_host call when nobody listens for events; same samples as the reference engine block by block."""
    from elementary_amd.offline import OfflineRenderer
    from elementary_amd.runtime import Runtime
    made = []

    def hip(sr, bs):
        made.append(Runtime(sr, bs, device=0))
        return made[-1]
    outs = []
    for factory in (hip, _checker):
        r = OfflineRenderer(factory)
        r.initialize(num_input_channels=1, num_output_channels=2, sample_rate=48000, block_size=512)
        r.render(el.mul(0.5, el.add(el.cycle(220.0), el.in_({"channel": 0}))), el.lowpass(900.0, 1.2, el.blepsaw(110.0)))
        x = lcg_noise(70 * 512 + 33, 5, 0.25)
        y = [np.zeros(70 * 512 + 33, dtype=np.float32) for _ in range(2)]
        r.process([x], y)
        outs.append(np.stack(y))
    assert made[0].stats()["batch_launches"] >= 1            # launch sets, not 71 single-block calls
    assert float(np.abs(outs[0] - outs[1]).max()) <= TOL


def test_cli_benchmark_host(gpu_required):
    """examples/benchmark_main.cpp (the timing protocol of cli/Benchmark.cpp:31-112 on the C++ facade), built by
    `make -C elementary_amd/csrc`, run on the C1 graph: its last block equals the reference engine's."""
    exe = os.path.join(ROOT, "examples", "bench_cli")
    if not os.path.exists(exe):
        res = subprocess.run(["make", "-C", os.path.join(ROOT, "elementary_amd", "csrc")], capture_output=True, text=True)
        assert res.returncode == 0, res.stderr[-2000:]
    sent = []
    Renderer(lambda b: sent.append(b) or 0).render(*graphs.c1_graph())
    c = _checker(graphs.C1_SAMPLE_RATE, 512)
    assert c.apply_instructions(sent[0]) == 0
    blocks = 300
    with tempfile.TemporaryDirectory() as d:
        bpath, opath = os.path.join(d, "batch.json"), os.path.join(d, "last.f32")
        open(bpath, "w").write(batch_to_json(sent[0]))
        res = subprocess.run([exe, bpath, str(blocks), str(graphs.C1_SAMPLE_RATE), opath], capture_output=True, text=True, timeout=300)
        assert res.returncode == 0, res.stderr
        assert "Average iteration time" in res.stdout
        got = np.fromfile(opath, dtype=np.float32).reshape(2, 512)
    for _ in range(blocks):                                   # warm-up block + (blocks - 1) timed ones precede the last
        c.process(None, 2, 512)
    ref = c.process(None, 2, 512)
    assert float(np.abs(got - ref).max()) <= TOL


@pytest.mark.parametrize("bs", [1024, 2048, 700, 1023, 521, 1031])
def test_host_blocks_longer_than_512_frames(gpu_required, bs):
    """Runtime(sr, blockSize > 512) (the reference has no limit, Runtime.h:44): a block that is a multiple of 512 frames is rendered
    as slices of 512. elemhip_process with full and short blocks and elemhip_process_blocks_host (whole HOST blocks: the state
    after a ragged call is the reference's) against the reference engine created with the same block size, on the 16-voice synth
    and on the every-stateful-node graph with a host input; the device-resident entry point answers 102."""
    from elementary_amd.runtime import Runtime, ElemHipError
    from cases import every_stateful_roots
    for roots_fn, n_out, n_in in ((lambda: graphs.c2_graph(voices=16), 2, 0), (every_stateful_roots, 3, 1)):
        a, c = Runtime(48000.0, bs, device=0), _checker(48000.0, bs)
        a.set_option("specialize", 2)
        assert a.render(*roots_fn())["result"] == 0 and c.render(*roots_fn())["result"] == 0
        worst, k = 0.0, 0
        # (r05: any size that splits into equal slices of 64 .. 512 frames: 700 -> 2 x 350, 1023 -> 3 x 341; a size nothing divides —
        #  521, 1031 are primes — as slices of 512 and a shorter last one)
        for n in (bs, bs, bs * 2 // 3 + 1, bs, min(512, bs - 1), bs):          # process(): full, short and one-slice calls
            x = np.stack([lcg_noise(n, 7 + k, 0.5)]) if n_in else None
            got, ref = a.process(x, n_out, n), c.process(x, n_out, n)
            assert got.shape == ref.shape == (n_out, n)
            worst = max(worst, float(np.abs(got - ref).max()) / max(1.0, float(np.abs(ref).max())))
            k += 1
        assert worst <= TOL, worst
        # the offline block loop over host arrays: 5 host blocks and a ragged sixth, then process() again (same state as the reference's)
        frames = 5 * bs + 333
        x = np.stack([lcg_noise(frames, 99, 0.5)]) if n_in else None
        got = a.process_blocks_host(x, n_out, frames)
        nb = (frames + bs - 1) // bs
        xp = np.zeros((1, nb * bs), dtype=np.float32)
        if n_in:
            xp[0, :frames] = x[0]
        ref = np.concatenate([c.process(xp[:, b * bs:(b + 1) * bs] if n_in else None, n_out, bs) for b in range(nb)], axis=1)[:, :frames]
        assert float(np.abs(got - ref).max()) <= TOL * max(1.0, float(np.abs(ref).max()))
        x = np.stack([lcg_noise(bs, 5, 0.5)]) if n_in else None
        got, ref = a.process(x, n_out, bs), c.process(x, n_out, bs)
        assert float(np.abs(got - ref).max()) <= TOL * max(1.0, float(np.abs(ref).max()))
        with pytest.raises(ElemHipError):
            a.process_blocks(2, n_out)
        if bs % 512 == 0:                      # (slices that are no multiple of 64 frames render through the interpreter kernels: plan.cpp specIsland)
            assert a.stats()["spec_launches"] > 0


@pytest.mark.parametrize("bs", [1024, 521])
def test_default_delay_sizes_follow_the_host_block(gpu_required, bs):
    """`delay` and `sdelay` created WITHOUT a size take the block size (Delays.h:56, 183): an `sdelay` then delays by one HOST block and
    a `delay` may reach that far back — the engine's slice (512 frames here) must not show through."""
    from elementary_amd.runtime import Runtime

    def roots():
        x = el.in_({"channel": 0})
        return [el.sdelay({}, x), el.delay({}, el.const({"value": float(bs - 3)}), 0.0, x)]
    a, c = Runtime(48000.0, bs, device=0), _checker(48000.0, bs)
    assert a.render(*roots())["result"] == 0 and c.render(*roots())["result"] == 0
    for k in range(6):
        x = np.stack([lcg_noise(bs, 11 + k, 0.5)])
        got, ref = a.process(x, 2, bs), c.process(x, 2, bs)
        assert float(np.abs(got - ref).max()) <= TOL, (bs, k)
        if k >= 2:
            assert float(np.abs(ref).max()) > 0.01          # (both delays have come through)
