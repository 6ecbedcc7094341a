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


from cases import NODE_CASES, sampleseq_scenario


@pytest.mark.parametrize("name", sorted(NODE_CASES))
def test_node_parity(gpu_required, name):
    roots_fn, n_in = NODE_CASES[name]
    hip, chk = _engines()
    a, b = render_pair(hip, chk, roots_fn, sample_rate=44100.0, blocks=14, n_in=n_in)
    assert np.isfinite(a).all()
    scale = max(1.0, float(np.abs(b).max()))
    err = float(np.abs(a - b).max())
    assert err <= TOL * scale, f"{name}: max abs err {err:.3e} (max |ref| {scale:.3g})"


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
    assert a.stats()["graph_replays"] > 0


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
