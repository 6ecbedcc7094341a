"""The C++ restatement (oracle/elem_oracle.cpp) against the unmodified reference engine
(oracle/_ref) on every node case and on the BASELINE graphs: same compiler, same libm, no FMA
contraction on either side, so the two must agree BIT FOR BIT."""
import numpy as np
import pytest

import oracle
from cases import NODE_CASES, REF_ONLY, node_case_resources, sampleseq_scenario
from elementary_amd import graphs
from helpers import render_pair

pytestmark = pytest.mark.skipif(not (oracle.have_ref() and oracle.have_port()), reason="needs oracle/_ref and the port")


def port(sr, bs):
    return oracle.PortRuntime(sr, bs)


def ref(sr, bs):
    return oracle.RefRuntime(sr, bs)


@pytest.mark.parametrize("name", sorted(set(NODE_CASES) - REF_ONLY))
def test_node_case_bit_exact(name):
    fn, n_in = NODE_CASES[name]
    a, b = render_pair(port, ref, fn, sample_rate=44100.0, blocks=14, n_in=n_in, resources=node_case_resources())
    assert np.array_equal(a, b), f"{name}: max abs diff {np.abs(a - b).max():.3e}"


def test_c1_and_c2_bit_exact():
    a, b = render_pair(port, ref, graphs.c1_graph, sample_rate=graphs.C1_SAMPLE_RATE, blocks=100)
    assert np.array_equal(a, b)
    a, b = render_pair(port, ref, lambda: graphs.c2_graph(64), sample_rate=graphs.C2_SAMPLE_RATE, blocks=60)
    assert np.array_equal(a, b)


def test_c4_instances_bit_exact():
    a, b = render_pair(port, ref, lambda: [graphs.c4_instance(k) for k in range(6)], sample_rate=graphs.C4_SAMPLE_RATE, blocks=60)
    assert np.array_equal(a, b)


def test_gc_matches_reference():
    """gc.test.js:5-43 pattern: nodes of a replaced graph are pruned only after the next rebuild."""
    from elementary_amd import el
    res = []
    for mk in (port, ref):
        rt = mk(44100.0, 512)
        rt.render(el.mul(2, 3))
        for _ in range(4):
            rt.process(None, 1, 512)
        first = rt.gc()
        rt.render(el.mul(3, 4))
        for _ in range(10):
            rt.process(None, 1, 512)
        second = rt.gc()
        rt.render(el.mul(4, 5))
        for _ in range(10):
            rt.process(None, 1, 512)
        third = rt.gc()
        res.append((sorted(first), sorted(second), sorted(third)))
    assert res[0] == res[1]
    assert res[0][0] == [] and len(res[0][2]) > 0


def test_sampleseq_scenario_bit_exact():
    """builtins/SampleSeq.h: restatement vs the reference engine over the sampleseq.test.js script."""
    a, b = sampleseq_scenario(port), sampleseq_scenario(ref)
    assert np.array_equal(a, b)
    assert np.abs(b).max() > 0.5
    a, b = sampleseq_scenario(port, block=512), sampleseq_scenario(ref, block=512)
    assert np.array_equal(a, b)


def test_event_relay_matches_reference():
    """Runtime::processQueuedEvents: same (type, payload) sequence from the restatement and the reference."""
    from elementary_amd import el
    from helpers import lcg_noise
    logs = []
    for mk in (port, ref):
        rt = mk(44100.0, 128)
        x = el.in_({"channel": 0})
        assert rt.render(el.meter({"name": "in"}, x), el.snapshot({"name": "snap"}, el.train(500.0), el.mul(2, x)),
                         el.meter({}, el.cycle(100.0)),
                         el.scope({"name": "sc", "size": 256, "channels": 2}, x, el.mul(0.5, x), el.mul(0.25, x)))["result"] == 0
        log = []
        for k in range(12):
            rt.process(lcg_noise(128, 5 + k, 0.5)[None, :], 4, 128)
            if k % 3 != 1:                      # skipped relays: only the newest readout survives
                log.append(rt.process_queued_events())
        logs.append(log)
    assert logs[0] == logs[1]
    assert sum(len(e) for e in logs[0]) > 16
