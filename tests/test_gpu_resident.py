"""Option ``resident``: ``elemhip_process`` through a kernel that stays on the GPU between calls (resident.hip).

The reference's ``Runtime::process`` (runtime/elem/Runtime.h:275-291) is one walk of the render sequence on the calling thread: no
launch, nothing to wait for. The launch path of the HIP engine pays two or more kernel launches and a stream synchronise per block;
the resident kernel takes blocks through a word in mapped host memory instead. What it renders has to be, bit for bit, what the
launch path renders (same island body, same records and arena) — through property changes, commits, event relays, an idle GPU and
graphs of several launch levels, each of which makes the kernel leave and come back.
"""
import time

import numpy as np
import pytest

from elementary_amd import el
from elementary_amd import graphs
from helpers import lcg_noise_fast

pytestmark = pytest.mark.gpu
TOL = 1e-6


def _hip(sr=44100.0, bs=512, **opts):
    from elementary_amd.runtime import Runtime
    rt = Runtime(sr, bs, device=0)
    for k, v in opts.items():
        rt.set_option(k, v)
    return rt


def _ref(sr=44100.0, bs=512):
    import oracle
    return oracle.RefRuntime(sr, bs) if oracle.have_ref() else oracle.PortRuntime(sr, bs)


def _run(rt, roots, blocks, n_in, n_out, between=None, bs=512):
    assert rt.render(*roots)["result"] == 0
    x = np.stack([lcg_noise_fast(blocks * bs, 5 + c, 0.5) for c in range(n_in)]) if n_in else None
    out = []
    for b in range(blocks):
        if between is not None:
            between(rt, b)
        out.append(rt.process(None if x is None else x[:, b * bs:(b + 1) * bs], n_out, bs).copy())
    return np.concatenate(out, axis=1)


def _stats(rt):
    return rt.stats() if hasattr(rt, "stats") else {}


def test_cli_benchmark_graph_resident_equals_launch_path(gpu_required):
    """The cli benchmark's graph (cli/Benchmark.cpp:31-112): 200 blocks, bit-identical to the launch path and within 1e-6 of the
    reference engine; one launch of the resident kernel renders all but the first few blocks."""
    a = _hip(resident=1, specialize=0)
    b = _hip(specialize=0)
    ya = _run(a, graphs.c1_graph(), 200, 0, 2)
    yb = _run(b, graphs.c1_graph(), 200, 0, 2)
    yc = _run(_ref(), graphs.c1_graph(), 200, 0, 2)
    assert np.array_equal(ya, yb)
    assert float(np.abs(ya - yc).max()) <= TOL
    st = _stats(a)
    # (one launch when the host never stays away for `resident_idle_us` = 2 ms; a scheduling hiccup of this Python loop on a shared box
    #  makes the kernel leave and come back — legal, seen once in r06: the bar is that the resident kernel rendered the run)
    assert 1 <= st["resident_launches"] <= 3 and st["resident_blocks"] >= 170 and st["blocks_rendered"] == 200
    assert _stats(b)["resident_blocks"] == 0


@pytest.mark.parametrize("spec", [0, 2])
def test_many_islands_and_levels_with_inputs(gpu_required, spec):
    """16 synth voices, a filtered input channel and a feedback pair: several islands per level, several levels (device-wide
    barriers between them), host input blocks brought in by the kernel, tap buffers promoted by it."""
    def roots():
        x = el.in_({"channel": 0})
        fb = el.tapOut({"name": "fb"}, el.mul(0.4, el.add(el.lowpass(700.0, 0.8, x), el.tapIn({"name": "fb"}))))
        v = graphs.c2_graph(16, 2)
        return [el.add(v[0], el.mul(0.5, fb)), el.add(v[1], el.highpass(300.0, 0.7, el.in_({"channel": 1})))]
    a = _hip(sr=48000.0, resident=1, specialize=spec)
    b = _hip(sr=48000.0, specialize=spec)
    ya, yb = _run(a, roots(), 120, 2, 2), _run(b, roots(), 120, 2, 2)
    yc = _run(_ref(48000.0), roots(), 120, 2, 2)
    st = _stats(a)
    assert st["num_levels"] >= 2 and st["num_islands"] >= 4, st
    assert st["resident_blocks"] >= 100
    assert float(np.abs(ya - yc).max()) <= TOL
    if spec == 0:
        assert np.array_equal(ya, yb)          # (specialised kernels and the interpreter body may differ in the last bit of a double)
    else:
        assert float(np.abs(ya - yb).max()) <= TOL


def test_property_changes_commits_and_event_relays_in_between(gpu_required):
    """Every 40th block a property is set (a keyed const: same graph, one SET_PROPERTY), every 70th the graph is rendered again with
    another filter (new nodes, a commit, root fades), every 55th the events are relayed: each time the kernel leaves, the launch
    path renders the blocks that have work to flush or fades running, and the kernel comes back."""
    def roots(gain, fc):
        x = el.in_({"channel": 0})
        g = el.const({"key": "g", "value": gain})
        return [el.mul(g, el.lowpass(fc, 0.7, x)), el.meter({"name": "m"}, el.mul(g, x))]

    def driver():
        state = {"gain": 0.5, "fc": 500.0, "events": []}

        def between(rt, blk):
            if blk and blk % 40 == 0:
                state["gain"] = 0.25 + 0.01 * (blk // 40)
                assert rt.render(*roots(state["gain"], state["fc"]))["result"] == 0
            if blk and blk % 70 == 0:
                state["fc"] = 500.0 + blk
                assert rt.render(*roots(state["gain"], state["fc"]))["result"] == 0
            if blk and blk % 55 == 0:
                state["events"].extend(rt.process_queued_events())
        return state, between
    a, b = _hip(resident=1), _hip()
    sa, fa = driver()
    sb, fb = driver()
    ya = _run(a, roots(0.5, 500.0), 300, 1, 2, fa)
    yb = _run(b, roots(0.5, 500.0), 300, 1, 2, fb)
    assert np.array_equal(ya, yb)
    assert len(sa["events"]) >= 5 and len(sa["events"]) == len(sb["events"])
    for ea, eb in zip(sa["events"], sb["events"]):
        assert ea == eb
    st = _stats(a)
    assert st["resident_launches"] >= 8 and st["resident_blocks"] >= 150, st


def test_idle_kernel_leaves_and_the_next_block_still_renders(gpu_required):
    """`resident_idle_us` = 300: a host that stays away for 20 ms finds the kernel gone (the GPU is free in between); the block goes
    through the launch path and the kernel is launched again after a few more."""
    a, b = _hip(resident=1, resident_idle_us=300), _hip()

    def nap(rt, blk):
        if blk in (50, 51, 90) and rt is a:
            time.sleep(0.02)
    ya = _run(a, graphs.c1_graph(), 140, 0, 2, nap)
    yb = _run(b, graphs.c1_graph(), 140, 0, 2)
    assert np.array_equal(ya, yb)
    st = _stats(a)
    assert st["resident_launches"] >= 3 and st["resident_blocks"] >= 100, st


def test_resident_call_latency_is_reported(gpu_required):
    """Not a benchmark (benchmarks/driver_configs.py c1 is) and not a win: measured on the cli benchmark's graph the call through
    the resident kernel takes 34 us from a native host against 24.5 us through the launch path (r05: the launch path's kernels take
    15 us of the call and launches + synchronise 9 us — profiles/r05/c1_sync_call_kernel_trace.txt — while the resident kernel renders
    with the interpreter island body, 27 us for this graph, behind device-wide barriers). The option stays opt-in; this test only
    pins that it is in the same league and prints both."""
    def timed(rt):
        assert rt.render(*graphs.c1_graph())["result"] == 0
        for _ in range(50):
            rt.process(None, 2, 512)
        lat = []
        for _ in range(1500):
            t0 = time.perf_counter()
            rt.process(None, 2, 512)
            lat.append(1e6 * (time.perf_counter() - t0))
        lat.sort()
        return lat[len(lat) // 2], lat[int(0.99 * len(lat))]
    r50, r99 = timed(_hip(resident=1))
    l50, l99 = timed(_hip())
    print(f"process() from Python, C1: resident p50 {r50:.1f} us p99 {r99:.1f} | launch path p50 {l50:.1f} us p99 {l99:.1f}")
    assert r50 <= 3.0 * l50 + 10.0


def test_sync_poll_equals_stream_synchronise(gpu_required):
    """`sync_poll` (default 1): elemhip_process returns when the block's epilogue kernel has published its word to mapped host
    memory behind the output block, instead of synchronising the stream (island.inc publish_done; C1 24.4 -> 19.6 us, C2 42 -> 36.6 us
    per call from a native host). Same samples bit for bit as `sync_poll` = 0, through fades (the per-block epilogue), settled
    blocks (the set-of-one epilogue), a property change, a re-render and an event relay; the counters say which wait was used."""
    def roots(gain, fc):
        x = el.in_({"channel": 0})
        g = el.const({"key": "g", "value": gain})
        return [el.mul(g, el.lowpass(fc, 0.7, x)), el.add(graphs.c2_graph(8, 2)[0], el.meter({"name": "m"}, el.mul(g, x)))]

    def between(rt, blk):
        if blk == 60:
            assert rt.render(*roots(0.3, 500.0))["result"] == 0
        if blk == 120:
            assert rt.render(*roots(0.3, 900.0))["result"] == 0
        if blk == 150:
            rt.process_queued_events()
    out = {}
    for poll in (1, 0):
        for spec in (0, 2):
            rt = _hip(sync_poll=poll, specialize=spec)
            out[poll, spec] = _run(rt, roots(0.5, 500.0), 220, 1, 2, between)
            d = rt.describe_plan()
            assert d["sync_poll"] == poll
            assert (d["sync_polls"] >= 200) if poll else (d["sync_polls"] == 0), d["sync_polls"]
            assert d["sync_poll_fallbacks"] == 0
    for spec in (0, 2):
        assert np.array_equal(out[1, spec], out[0, spec])
    ref = _run(_ref(), roots(0.5, 500.0), 220, 1, 2, between)
    assert float(np.abs(out[1, 2] - ref).max()) <= TOL
