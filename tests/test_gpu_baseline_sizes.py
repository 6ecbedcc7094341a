"""Parity at the sizes BASELINE.json states for its configs, and the dynamic-graph stream of config 5:
every output block and every gc() result of the HIP engine vs the reference engine. Tolerance 1e-6 absolute."""
import numpy as np
import pytest

from elementary_amd import el, graphs

pytestmark = pytest.mark.gpu
TOL = 1e-6


def _checker(sr, bs):
    import oracle
    return oracle.RefRuntime(sr, bs) if oracle.have_ref() else oracle.PortRuntime(sr, bs)


def _hip(sr, bs, specialize=0, **opts):
    from elementary_amd.runtime import Runtime
    rt = Runtime(sr, bs, device=0)
    rt.set_option("specialize", specialize)
    for k, v in opts.items():
        rt.set_option(k, v)
    return rt


def _blocks(rt, nb, n_out, block=512):
    import torch
    out = torch.zeros((nb, n_out, block), dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    rt.process_blocks(nb, n_out, out_ptr=out.data_ptr())
    return out.cpu().numpy()


def _voices_graph(ids, channels=2):
    """The C2 mix (graphs.c2_graph) over an explicit list of voice ids."""
    outs = []
    for c in range(channels):
        vs = [graphs.c2_voice(v) for k, v in enumerate(ids) if k % channels == c]
        outs.append(el.add(*vs) if len(vs) > 1 else vs[0])
    return outs


def test_gc_timing_on_a_rendering_engine(gpu_required):
    """gc.test.js:5-47 on the HIP engine WHILE IT RENDERS: the device-side root fades (mirrored on the host) decide when
    the first graph's nodes become collectable — same blocks, same gc() results as the reference engine."""
    logs = []
    for mk in (lambda: _hip(44100.0, 512), lambda: _checker(44100.0, 512)):
        rt = mk()
        log = []
        assert rt.render(el.mul(2, 3))["result"] == 0
        early = sorted(rt.renderer._delegate.node_map.keys())
        ys = [rt.process(None, 1, 512) for _ in range(10)]
        log.append(("gc0", sorted(rt.gc())))
        assert rt.render(el.mul(4, 5))["result"] == 0
        ys += [rt.process(None, 1, 512) for _ in range(10)]
        log.append(("gc1", sorted(rt.gc())))
        assert rt.render(el.mul(6, 7))["result"] == 0
        ys += [rt.process(None, 1, 512) for _ in range(10)]
        pruned = sorted(rt.gc())
        log.append(("gc2", pruned))
        assert all(k not in rt.renderer._delegate.node_map for k in pruned)
        logs.append((log, np.stack(ys), early))
    (la, ya, early_a), (lb, yb, early_b) = logs
    assert la == lb and early_a == early_b
    assert lb[0][1] == [] and lb[1][1] == [] and lb[2][1] == early_b      # gc.test.js expectations
    assert float(np.abs(ya - yb).max()) <= TOL


@pytest.mark.parametrize("path", ["process", "process_blocks", "small_heap"])
def test_c5_mutation_stream(gpu_required, path):
    """BASELINE configs[4]: a live 128-voice graph (2060 nodes), one voice replaced per batch the way the reconciler does
    it (new voice nodes, a new mix add and a new root per channel; the old root fades out over 20 ms), gc() every 16
    batches. Every block and every pruned-id set must equal the reference engine's."""
    a, c = _hip(graphs.C2_SAMPLE_RATE, 512, batch_blocks=4), _checker(graphs.C2_SAMPLE_RATE, 512)
    if path == "small_heap":      # the device heap of island programs runs out every few re-plans: a fresh heap, everything uploaded again,
        a.set_option("prog_heap_dwords", 290_000)   # while the blocks in flight still read the old one
    ids = list(range(128))
    nxt = 128
    worst = 0.0
    pruned_total = 0
    for batch in range(41):
        if batch:
            ids[(batch * 37) % 128] = nxt
            nxt += 1
        roots = _voices_graph(ids)
        ra, rc = a.render(*roots), c.render(*roots)
        assert ra["result"] == 0 and rc["result"] == 0
        assert ra["batch"] == rc["batch"]
        nb = 1 + batch % 3                                   # 10-32 ms between batches: fades overlap the next mutation
        ref = np.stack([c.process(None, 2, 512) for _ in range(nb)])
        got = _blocks(a, nb, 2) if path == "process_blocks" else np.stack([a.process(None, 2, 512) for _ in range(nb)])
        worst = max(worst, float(np.abs(got - ref).max()))
        assert worst <= TOL, f"batch {batch}: {worst:.3e}"
        if batch % 16 == 15:
            pa, pc = sorted(a.gc()), sorted(c.gc())
            assert pa == pc, (batch, len(pa), len(pc))
            pruned_total += len(pa)
    assert pruned_total > 100                                # replaced voices, old mix adds and old roots were reclaimed
    heaps = a.describe_plan()["plan_prog_heaps"]
    assert heaps >= 4 if path == "small_heap" else heaps == 1
    # the stream settles: a last long stretch, then everything that is not in the live graph goes
    ref = np.stack([c.process(None, 2, 512) for _ in range(8)])
    got = _blocks(a, 8, 2) if path == "process_blocks" else np.stack([a.process(None, 2, 512) for _ in range(8)])
    assert float(np.abs(got - ref).max()) <= TOL
    assert sorted(a.gc()) == sorted(c.gc())


@pytest.mark.parametrize("specialize", [0, 2])
def test_c4_full_size(gpu_required, specialize):
    """BASELINE configs[3], one GPU's share: 128 independent render instances x 375 blocks (4 s of audio), through the
    offline entry point; the delay{size:24000} rings wrap 8 times."""
    a, c = _hip(graphs.C4_SAMPLE_RATE, 512, specialize=specialize), _checker(graphs.C4_SAMPLE_RATE, 512)
    roots = [graphs.c4_instance(k) for k in range(128)]
    assert a.render(*roots)["result"] == 0 and c.render(*roots)["result"] == 0
    got = _blocks(a, 375, 128)
    ref = np.stack([c.process(None, 128, 512) for _ in range(375)])
    if specialize:
        assert a.stats()["spec_launches"] > 0
    assert np.abs(ref).max() > 0.05
    assert float(np.abs(got - ref).max()) <= TOL


@pytest.mark.parametrize("specialize", [0, 2])
def test_c2_soak_600_blocks(gpu_required, specialize):
    """BASELINE configs[1] for 600 blocks (6.4 s): long enough for every envelope and oscillator phase to drift if a
    recurrence were off by an ulp per block."""
    a, c = _hip(graphs.C2_SAMPLE_RATE, 512, specialize=specialize), _checker(graphs.C2_SAMPLE_RATE, 512)
    roots = graphs.c2_graph()
    assert a.render(*roots)["result"] == 0 and c.render(*roots)["result"] == 0
    got = np.concatenate([_blocks(a, 200, 2) for _ in range(3)])
    ref = np.stack([c.process(None, 2, 512) for _ in range(600)])
    err = float(np.abs(got - ref).max())
    assert err <= TOL, f"{err:.3e}"


def test_rendering_continues_while_a_commit_is_planned(gpu_required):
    """The render thread keeps producing blocks of the CURRENT sequence while another thread's applyInstructions builds the
    next plan (the reference's SPSC hand-over, Runtime.h:207-216 / 277-285). With the build stretched to 0.6 s the stream
    must be what the reference's audio thread would see: the instructions in front of COMMIT_UPDATES take effect at once
    (the replaced root starts its fade-out on the old sequence, Runtime.h:368-433), the new sequence arrives later and
    its root fades in from there. The reference engine is driven through the same timeline single-threaded: the batch
    without its commit first, `[ACTIVATE_ROOTS (same ids), COMMIT_UPDATES]` at the block where the HIP stream switches
    (two reference engines in lock-step find that block). No process call made meanwhile waits for the build."""
    import threading, time
    from elementary_amd.runtime import Runtime
    sr = 48000.0
    old = [el.mul(0.5, el.cycle(220.0)), el.mul(0.25, el.cycle(331.0))]
    new = [el.mul(0.5, el.cycle(220.0)), el.mul(0.3, graphs.c2_voice(3))]
    a, c1, c2 = _hip(sr, 512), _checker(sr, 512), _checker(sr, 512)
    scribe = Runtime(sr, 512, device=-1)          # same reconciler history: yields the batch `a.render(*new)` will send
    for rt in (a, c1, c2, scribe):
        assert rt.render(*old)["result"] == 0
    batch = scribe.render(*new)["batch"]
    assert batch[-1] == [5] and batch[-2][0] == 4
    early, late = batch[:-1], [batch[-2], [5]]
    for _ in range(4):
        got, ref = a.process(None, 2, 512), c1.process(None, 2, 512)
        c2.process(None, 2, 512)
        assert float(np.abs(got - ref).max()) <= TOL
    a.set_option("debug_build_delay_ms", 600)
    done = {}

    def commit():
        done["rc"] = a.render(*new)["result"]

    th = threading.Thread(target=commit)
    th.start()
    time.sleep(0.15)                              # the commit is inside its (unlocked) build: its early instructions are in
    assert c1.apply_instructions(early) == 0 and c2.apply_instructions(early) == 0
    import gc
    gc.collect(); gc.disable()      # (a full collection of the test session's heap takes longer than the latency bound below)
    worst, during, switched = 0.0, 0, False
    while during < 100000:                        # (~80 us per block: the 0.6 s build spans several thousand blocks)
        t0 = time.perf_counter()
        got = a.process(None, 2, 512)
        dt = time.perf_counter() - t0
        if float(np.abs(got - c1.process(None, 2, 512)).max()) <= TOL:      # still the old sequence
            c2.process(None, 2, 512)
            worst = max(worst, dt)
            during += 1
            continue
        assert c2.apply_instructions(late) == 0                             # the new sequence arrived in front of this block
        assert float(np.abs(got - c2.process(None, 2, 512)).max()) <= TOL
        switched = True
        break
    th.join()
    gc.enable()
    assert done["rc"] == 0 and switched
    assert during >= 20 and worst < 0.1, (during, worst)
    for k in range(12):      # the new root's fade-in, then the settled new graph
        got, ref = a.process(None, 2, 512), c2.process(None, 2, 512)
        assert float(np.abs(got - ref).max()) <= TOL, k


@pytest.mark.parametrize("batch,rows,split,spec", [(1024, 64, 2, 2), (256, 64, 2, 2), (128, 16, 1, 0), (64, 8, 8, 2), (7, 3, 4, 0)])
def test_launch_set_geometry_options(gpu_required, batch, rows, split, spec):
    """Blocks per launch set (up to 1024), grid rows of stateless islands and workgroups per mixer only change how the
    work is laid out: 300 blocks of a 48-voice C2 graph equal the reference engine's for every setting."""
    a, c = _hip(graphs.C2_SAMPLE_RATE, 512, specialize=spec, batch_blocks=batch, stateless_rows=rows, mixer_split=split), _checker(graphs.C2_SAMPLE_RATE, 512)
    roots = graphs.c2_graph(voices=48)
    assert a.render(*roots)["result"] == 0 and c.render(*roots)["result"] == 0
    got = _blocks(a, 300, 2)
    ref = np.stack([c.process(None, 2, 512) for _ in range(300)])
    assert a.stats()["batch_launches"] >= max(1, 300 // batch)
    assert float(np.abs(got - ref).max()) <= TOL


@pytest.mark.parametrize("path", ["process", "process_blocks"])
def test_idle_launches_of_a_replaced_root_are_left_out(gpu_required, path):
    """A replaced root stays in the plan until the next commit (GraphRenderSequence.h:214-219: it renders while its fade-out
    runs, then not at all). Once its fade has settled the launch that would only start its islands is left out by the host
    (launchLevelBatch: mirror of the root fades) — every block still equals the reference engine's, also when the number of
    output channels the caller asks for changes what runs."""
    a, c = _hip(graphs.C2_SAMPLE_RATE, 512, batch_blocks=4), _checker(graphs.C2_SAMPLE_RATE, 512)
    a.set_option("specialize", 2)       # (the replaced voice's one-off island shape gets a kernel of its own: ~10 s of compiles on a cold cache)
    ids = list(range(32))
    worst, nxt = 0.0, 32
    for batch in range(6):
        if batch:
            ids[(batch * 11) % 32] = nxt
            nxt += 1
        roots = _voices_graph(ids)
        assert a.render(*roots)["result"] == 0 and c.render(*roots)["result"] == 0
        for k in range(4):
            n_out = 1 if (batch == 3 and k == 2) else 2                      # one call that leaves channel 1 out
            nb = 1 if path == "process" else 5
            ref = np.stack([c.process(None, n_out, 512) for _ in range(nb)])
            got = np.stack([a.process(None, n_out, 512) for _ in range(nb)]) if path == "process" else _blocks(a, nb, n_out)
            worst = max(worst, float(np.abs(got - ref).max()))
            assert worst <= TOL, (batch, k, worst)
    info = a.describe_plan()
    assert info["plan_idle_launches_skipped"] > 0
    assert info["plan_spec_fade_blocks"] >= 5          # the blocks right after a commit: specialised level launches + the per-block epilogue
