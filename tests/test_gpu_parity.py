"""GPU parity: the HIP engine (through the C-ABI) vs the reference engine on the same instruction
batches and inputs.  Tolerance 1e-6 absolute (BASELINE.json north_star); float-state recurrences
are expected to be bit-exact, transcendental nodes differ by libm ulps only."""
import math

import numpy as np
import pytest

from elementary_amd import el, graphs
from helpers import lcg_noise, render_pair

pytestmark = pytest.mark.gpu

TOL = 1e-6


def _engines():
    import oracle
    from elementary_amd.runtime import Runtime

    def hip(sr, bs):
        return Runtime(sr, bs, device=0)

    if oracle.have_ref():
        def chk(sr, bs):
            return oracle.RefRuntime(sr, bs)
    else:
        def chk(sr, bs):
            return oracle.PortRuntime(sr, bs)
    return hip, chk


from cases import NODE_CASES, REF_ONLY, node_case_resources, sampleseq_scenario


@pytest.mark.parametrize("name", sorted(NODE_CASES))
def test_node_parity(gpu_required, name):
    roots_fn, n_in = NODE_CASES[name]
    if name in REF_ONLY:
        import oracle
        if not oracle.have_ref():
            pytest.skip("needs oracle/_ref")
    hip, chk = _engines()
    a, b = render_pair(hip, chk, roots_fn, sample_rate=44100.0, blocks=14, n_in=n_in, resources=node_case_resources())
    assert np.isfinite(a).all()
    scale = max(1.0, float(np.abs(b).max()))
    err = float(np.abs(a - b).max())
    assert err <= TOL * scale, f"{name}: max abs err {err:.3e} (max |ref| {scale:.3g})"


@pytest.mark.parametrize("name", sorted(NODE_CASES))
def test_node_batched_equals_blockwise(gpu_required, name):
    """Every node case through multi-block launches (5 blocks per launch, 17 blocks: three full launches and a ragged one)
    vs the same engine block by block: bit-identical. Small graphs give every recurrence a wave of its own, so this is
    the coverage of the tasks that render a whole launch without returning to the walk (kTaskOwnsWave)."""
    import torch
    from elementary_amd.runtime import Runtime
    roots_fn, n_in = NODE_CASES[name]
    nb = 17
    a, b = Runtime(44100.0, 512), Runtime(44100.0, 512)
    a.set_option("batch_blocks", 5)
    for rt in (a, b):
        for rname, data in node_case_resources().items():
            assert rt.add_shared_resource(rname, data)
    roots = roots_fn()
    n_out = len(roots)
    assert a.render(*roots)["result"] == 0 and b.render(*roots)["result"] == 0
    x = np.stack([np.stack([lcg_noise(512, 1 + c + 97 * k, 0.5) for c in range(max(n_in, 1))]) for k in range(nb)])
    xin = torch.from_numpy(x).cuda()
    out = torch.zeros((nb, n_out, 512), dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    a.process_blocks(nb, n_out, out_ptr=out.data_ptr(), in_ptr=xin.data_ptr(), num_inputs=max(n_in, 1))
    ref = np.stack([b.process(x[k], n_out, 512) for k in range(nb)])
    got = out.cpu().numpy()
    assert np.array_equal(got, ref), f"{name}: max abs diff {np.abs(got - ref).max():.3e}"


def test_c1_benchmark_graph(gpu_required):
    hip, chk = _engines()
    a, b = render_pair(hip, chk, graphs.c1_graph, sample_rate=graphs.C1_SAMPLE_RATE, blocks=200)
    assert float(np.abs(a - b).max()) <= TOL


def test_c2_full_graph_200_blocks(gpu_required):
    """BASELINE configs[1] at full size: 4107 nodes, parity over the first 200 blocks."""
    hip, chk = _engines()
    a, b = render_pair(hip, chk, graphs.c2_graph, sample_rate=graphs.C2_SAMPLE_RATE, blocks=200)
    err = float(np.abs(a - b).max())
    assert err <= TOL, f"C2 max abs err {err:.3e}, max|ref| {np.abs(b).max():.3f}"


def test_c4_instances(gpu_required):
    hip, chk = _engines()
    a, b = render_pair(hip, chk, lambda: [graphs.c4_instance(k) for k in range(8)], sample_rate=graphs.C4_SAMPLE_RATE, blocks=60)
    assert float(np.abs(a - b).max()) <= TOL


def test_short_blocks_and_odd_sizes(gpu_required):
    hip, chk = _engines()
    a, b = hip(44100.0, 512), chk(44100.0, 512)
    roots = [el.lowpass(800, 1.0, el.blepsaw(110.0)), el.delay({"size": 300}, 100.5, 0.5, el.cycle(50.0))]
    assert a.render(*roots)["result"] == 0 and b.render(*roots)["result"] == 0
    for n in [512, 1, 7, 64, 65, 500, 512, 33, 128]:
        ya, yb = a.process(None, 2, n), b.process(None, 2, n)
        assert ya.shape == (2, n)
        assert float(np.abs(ya - yb).max()) <= TOL, n


def test_process_blocks_matches_process(gpu_required):
    """Offline multi-block path (hipGraph replay, HBM-resident output) == block-by-block process()."""
    import torch
    from elementary_amd.runtime import Runtime
    roots = graphs.c2_graph(voices=16)
    a, b = Runtime(48000.0, 512), Runtime(48000.0, 512)
    assert a.render(*roots)["result"] == 0 and b.render(*roots)["result"] == 0
    nb = 37
    out = torch.zeros((nb, 2, 512), dtype=torch.float32, device="cuda")
    a.process_blocks(nb, 2, out_ptr=out.data_ptr())
    ref = np.stack([b.process(None, 2, 512) for _ in range(nb)])
    assert np.array_equal(out.cpu().numpy(), ref)
    # and again: state carried across calls, graph replay reused
    a.process_blocks(nb, 2, out_ptr=out.data_ptr())
    ref = np.stack([b.process(None, 2, 512) for _ in range(nb)])
    assert np.array_equal(out.cpu().numpy(), ref)
    assert a.stats()["batch_launches"] > 0   # multi-block pipelined launches carried the steady-state blocks
    # the per-block hipGraph path (batching off) renders the same samples
    c = Runtime(48000.0, 512)
    c.set_option("batch_blocks", 1)
    c.set_option("specialize", 0)            # (with specialised kernels loaded, single blocks go through them as sets of one, not through the captured graph)
    assert c.render(*roots)["result"] == 0
    c.process_blocks(nb, 2, out_ptr=out.data_ptr())
    d = Runtime(48000.0, 512)
    assert d.render(*roots)["result"] == 0
    ref = np.stack([d.process(None, 2, 512) for _ in range(nb)])
    assert np.array_equal(out.cpu().numpy(), ref)
    assert c.stats()["graph_replays"] > 0 and c.stats()["batch_launches"] == 0


@pytest.mark.parametrize("copies", [1, 2, 3, 4, 5])
def test_pipelined_batches_match_block_at_a_time(gpu_required, copies):
    """Multi-block launches (blocks pipelined through `copies` LDS buffer sets) vs process(): every
    stateful node type in one graph, host inputs, time-dependent nodes, 3 batches + a ragged tail."""
    import torch
    from elementary_amd.runtime import Runtime
    X = el.in_({"channel": 0})
    def roots():
        v = el.lowpass(el.add(900, el.mul(700, el.cycle(2.0))), 1.5, el.add(el.blepsaw(110.0), el.mul(0.5, X)))
        w = el.delay({"size": 3000}, el.add(1000.5, el.mul(300, el.cycle(0.5))), 0.4, el.pole(0.95, X))
        z = el.mul(el.adsr(0.002, 0.01, 0.5, 0.02, el.train(9.0)), el.pinknoise({"seed": 3}))
        t = el.add(el.mul(1e-5, el.time()), el.metro({"interval": 7.0}), el.sdelay({"size": 700}, X), el.z(X))
        s = el.add(el.biquad(0.2, 0.3, 0.2, -0.5, 0.2, X), el.mm1p({"mode": "lowpass"}, el.prewarp(800.0), X),
                   el.env(el.tau2pole(0.001), el.tau2pole(0.05), X), el.latch(el.train(60.0), X),
                   el.seq({"seq": [1, 2, 3, 5.5], "hold": True}, el.train(200.0), 0), el.counter(el.train(50.0)),
                   el.maxhold({"hold": 3.0}, el.abs(X), el.train(9.0)), el.accum(el.abs(X), el.train(20.0)),
                   el.highshelf(4000, 0.7, -4.5, X), el.syncphasor(440.0, el.train(37.0)), el.bleptriangle(523.25))
        return [el.tanh(el.add(v, w)), el.add(z, t), s]
    nb = 53
    x = np.stack([np.stack([lcg_noise(512, 7 + k, 0.5)]) for k in range(nb)])          # [nb, 1, 512]
    a, b = Runtime(48000.0, 512), Runtime(48000.0, 512)
    a.set_option("pipeline_copies", copies)
    a.set_option("batch_blocks", 12)
    assert a.render(*roots())["result"] == 0 and b.render(*roots())["result"] == 0
    xin = torch.from_numpy(x).cuda()
    out = torch.zeros((nb, 3, 512), dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    a.process_blocks(nb, 3, out_ptr=out.data_ptr(), in_ptr=xin.data_ptr(), num_inputs=1)
    ref = np.stack([b.process(x[k], 3, 512) for k in range(nb)])
    got = out.cpu().numpy()
    assert a.stats()["batch_launches"] >= 3
    assert np.array_equal(got, ref), float(np.abs(got - ref).max())


def test_property_update_without_rebuild(gpu_required):
    """ref.test.js:5-37 pattern: a keyed const changes value through SET_PROPERTY only."""
    hip, chk = _engines()
    for mk in (hip, chk):
        rt = mk(44100.0, 512)
        node, setter = rt.renderer.create_ref("const", {"value": 500}, [])
        assert rt.render(el.mul(0.001, node))["result"] == 0
        for _ in range(10):
            y = rt.process(None, 1, 512)
        assert np.allclose(y, 0.5)
        assert setter({"value": 800}) == 0
        y = rt.process(None, 1, 512)
        assert np.allclose(y, 0.8), y[0, :4]


def test_root_fade_and_switch_back(gpu_required):
    """offline-renderer.test.js:48-76: cross-fade to a new graph and back (root re-activation)."""
    hip, chk = _engines()
    outs = []
    for mk in (hip, chk):
        rt = mk(44100.0, 512)
        ys = []
        assert rt.render(el.mul(2, 3))["result"] == 0
        ys += [rt.process(None, 1, 512) for _ in range(4)]
        assert rt.render(el.mul(3, 4))["result"] == 0
        ys += [rt.process(None, 1, 512) for _ in range(4)]
        assert rt.render(el.mul(2, 3))["result"] == 0
        ys += [rt.process(None, 1, 512) for _ in range(6)]
        outs.append(np.stack(ys))
    assert float(np.abs(outs[0] - outs[1]).max()) <= TOL
    assert np.allclose(outs[1][-1], 6.0)


@pytest.mark.parametrize("block", [32, 512])
def test_sampleseq_scenario(gpu_required, block):
    """sampleseq.test.js:5-76 script (onset/offset fades, jumps, seq/buffer/duration swaps)."""
    hip, chk = _engines()
    a, b = sampleseq_scenario(hip, block=block), sampleseq_scenario(chk, block=block)
    assert np.abs(b).max() > 0.5
    assert float(np.abs(a - b).max()) <= TOL


def test_event_relay(gpu_required):
    """Runtime::processQueuedEvents (Runtime.h:437-446): meter / snapshot readouts of the HIP engine vs the checker."""
    hip, chk = _engines()
    logs = []
    for mk in (hip, chk):
        rt = mk(44100.0, 128)
        x = el.in_({"channel": 0})
        assert rt.render(el.meter({"name": "in"}, x), el.snapshot({"name": "snap"}, el.train(500.0), el.mul(2, x)),
                         el.meter({}, el.cycle(100.0)),
                         el.scope({"name": "sc", "size": 256, "channels": 2}, x, el.mul(0.5, x), el.mul(0.25, x)))["result"] == 0
        log = []
        for k in range(12):
            rt.process(lcg_noise(128, 5 + k, 0.5)[None, :], 4, 128)
            if k % 3 != 1:
                log.append(rt.process_queued_events())
        logs.append(log)
    assert len(logs[0]) == len(logs[1])
    for a, b in zip(*logs):
        assert [(t, p.get("source")) for t, p in a] == [(t, p.get("source")) for t, p in b]
        for (_, pa), (_, pb) in zip(a, b):
            for key in ("min", "max", "data"):
                if key in pb:
                    assert float(np.abs(np.asarray(pa[key], np.float64) - np.asarray(pb[key], np.float64)).max()) <= TOL, (key, pa, pb)


def test_mc_capture_events(gpu_required):
    """builtins/mc/Capture.h: the "mc.capture" relay of the HIP engine vs the reference engine — a slow gate (takes spanning
    several blocks), a noisy gate (many rises and falls inside one block: the writeStart / writeStop marker logic), events
    polled at irregular intervals, a re-render in between (the reference re-creates the node's ring on every push)."""
    import oracle
    if not oracle.have_ref():
        pytest.skip("needs oracle/_ref")
    from elementary_amd.runtime import Runtime
    logs, outs = [], []
    for mk in (lambda sr, bs: Runtime(sr, bs, device=0), lambda sr, bs: oracle.RefRuntime(sr, bs)):
        rt = mk(44100.0, 512)
        x0, x1 = el.in_({"channel": 0}), el.in_({"channel": 1})
        def graph(extra):
            return (el.mc.capture({"name": "slow", "channels": 2}, el.train(7.0), x0, el.mul(0.5, x1), el.mul(extra, x0))
                    + el.mc.capture({"name": "noisy", "channels": 1}, el.ge(x1, 0.2), x0, x1))
        assert rt.render(*graph(0.25))["result"] == 0
        log, out = [], []
        for k in range(40):
            x = np.stack([lcg_noise(512, 5 + k, 0.5), lcg_noise(512, 105 + k, 0.5)])
            out.append(rt.process(x, 3, 512))
            if k == 20:
                assert rt.render(*graph(0.75))["result"] == 0
            if k % 4 != 2:
                log.append(rt.process_queued_events())
        logs.append(log); outs.append(np.stack(out))
    assert float(np.abs(outs[0] - outs[1]).max()) <= TOL
    assert len(logs[0]) == len(logs[1])
    n_events = 0
    for a, b in zip(*logs):
        assert [(t, p.get("source")) for t, p in a] == [(t, p.get("source")) for t, p in b], (a, b)
        for (_, pa), (_, pb) in zip(a, b):
            assert len(pa["data"]) == len(pb["data"])
            for ca, cb in zip(pa["data"], pb["data"]):
                assert len(ca) == len(cb), (len(ca), len(cb))
                if len(cb):
                    assert float(np.abs(np.asarray(ca, np.float64) - np.asarray(cb, np.float64)).max()) <= TOL
                n_events += 1
    assert n_events >= 10


def test_mc_capture_gains_a_channel(gpu_required):
    """a LIVE mc.capture node whose second output channel is consumed only by a later render. The new
    channel's record used to be cloned from channel 0's device record, CAP_CH included: it passed input 1 through instead of
    input 2 and recorded every block a second time into the shared ring (duplicated takes)."""
    import oracle
    if not oracle.have_ref():
        pytest.skip("needs oracle/_ref")
    from elementary_amd.runtime import Runtime
    logs, outs = [], []
    for mk in (lambda sr, bs: Runtime(sr, bs, device=0), lambda sr, bs: oracle.RefRuntime(sr, bs)):
        rt = mk(44100.0, 512)
        x0, x1 = el.in_({"channel": 0}), el.in_({"channel": 1})
        cap = el.mc.capture({"name": "take", "channels": 2}, el.train(9.0), x0, el.mul(0.5, x1))
        assert rt.render(cap[0])["result"] == 0
        log, out = [], []
        for k in range(36):
            x = np.stack([lcg_noise(512, 7 + k, 0.5), lcg_noise(512, 207 + k, 0.5)])
            y = rt.process(x, 2, 512)
            out.append(y)
            if k == 11:
                assert rt.render(cap[0], cap[1])["result"] == 0     # the same node, one more consumed channel
            log.append(rt.process_queued_events())
        logs.append(log); outs.append(np.stack(out))
    assert float(np.abs(outs[0] - outs[1]).max()) <= TOL
    assert float(np.abs(outs[1][20:, 1]).max()) > 0.01                  # channel 1 really renders input 2 after the second render
    n_events = 0
    for a, b in zip(*logs):
        assert [(t, p.get("source")) for t, p in a] == [(t, p.get("source")) for t, p in b], (a, b)
        for (_, pa), (_, pb) in zip(a, b):
            assert [len(c) for c in pa["data"]] == [len(c) for c in pb["data"]]
            for ca, cb in zip(pa["data"], pb["data"]):
                if len(cb):
                    assert float(np.abs(np.asarray(ca, np.float64) - np.asarray(cb, np.float64)).max()) <= TOL
            n_events += 1
    assert n_events >= 4


def test_stranger_things_example(gpu_required):
    """cli/examples/02_StrangerThings.js:10-31 (the reference's own example patch, also the 69-node voice of
    hashing.test.js) rendered on both engines, two channels, 3 s of audio at the cli's sample rate."""
    from test_reconciler import stranger_things_voice
    hip, chk = _engines()
    a, b = render_pair(hip, chk, lambda: [stranger_things_voice(), stranger_things_voice()], sample_rate=44100.0, blocks=260)
    assert np.abs(b).max() > 0.05
    scale = max(1.0, float(np.abs(b).max()))
    assert float(np.abs(a - b).max()) <= TOL * scale


@pytest.mark.parametrize("bs", [37, 100, 192, 300, 448, 500])
def test_block_sizes_with_mixers(gpu_required, bs):
    """Block sizes that are not powers of two, with mixer islands in the plan (16 C2 voices: two 8-input mixers, cut into
    64-frame runs over the waves of one or more workgroups): block at a time and through launch sets vs the reference."""
    import torch
    hip, chk = _engines()
    a, b, c = hip(48000.0, bs), hip(48000.0, bs), chk(48000.0, bs)
    roots = graphs.c2_graph(voices=16)
    for rt in (a, b, c):
        assert rt.render(*roots)["result"] == 0
    nb = 24
    ref = np.stack([c.process(None, 2, bs) for _ in range(nb)])
    one = np.stack([a.process(None, 2, bs) for _ in range(nb)])
    out = torch.zeros((nb, 2, bs), dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    b.set_option("batch_blocks", 7)
    b.process_blocks(nb, 2, out_ptr=out.data_ptr())
    assert float(np.abs(one - ref).max()) <= TOL and float(np.abs(out.cpu().numpy() - ref).max()) <= TOL
