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


X = lambda ch=0: el.in_({"channel": ch})  # noqa: E731
POS = lambda ch=0: el.add(0.6, X(ch))     # noqa: E731  in [0.1, 1.1]

NODE_CASES = {
    # name: (roots_fn, n_in)
    "const_mul": (lambda: [el.mul(0.5, 2.0)], 0),
    "sr": (lambda: [el.div(el.sr(), 1000.0)], 0),
    "in_passthrough": (lambda: [X(0), X(1)], 2),
    "in_missing_channel": (lambda: [X(5)], 2),
    "sin": (lambda: [el.sin(el.mul(6.0, X()))], 1),
    "cos": (lambda: [el.cos(el.mul(6.0, X()))], 1),
    "tan": (lambda: [el.tan(X())], 1),
    "tanh": (lambda: [el.tanh(el.mul(4.0, X()))], 1),
    "asinh": (lambda: [el.asinh(el.mul(10.0, X()))], 1),
    "ln": (lambda: [el.ln(POS())], 1),
    "log": (lambda: [el.log(POS())], 1),
    "log2": (lambda: [el.log2(POS())], 1),
    "ceil_floor_round": (lambda: [el.ceil(el.mul(9.0, X())), el.floor(el.mul(9.0, X())), el.round(el.mul(9.0, X()))], 1),
    "sqrt": (lambda: [el.sqrt(POS())], 1),
    "exp": (lambda: [el.exp(el.mul(3.0, X()))], 1),
    "abs": (lambda: [el.abs(X())], 1),
    "compare": (lambda: [el.le(X(0), X(1)), el.leq(X(0), X(1)), el.ge(X(0), X(1)), el.geq(X(0), X(1))], 2),
    "pow": (lambda: [el.pow(POS(0), el.mul(3.0, X(1))), el.pow(X(0), 2.0), el.pow(X(0), 0.5)], 2),
    "eq_and_or": (lambda: [el.eq(el.round(el.mul(2, X(0))), el.round(el.mul(2, X(1)))),
                           el.and_(el.ge(X(0), 0), el.ge(X(1), 0)), el.or_(el.ge(X(0), 0), el.ge(X(1), 0))], 2),
    "reduce": (lambda: [el.add(X(0), X(1), 0.25, X(0)), el.sub(X(0), X(1), 0.1), el.mul(X(0), X(1), 3.0),
                        el.div(X(0), X(1)), el.div(X(0), el.floor(X(1)))], 2),
    "mod_min_max": (lambda: [el.mod(X(0), 0.3), el.min(X(0), X(1), 0.2), el.max(X(0), X(1), -0.2)], 2),
    "add_100": (lambda: [el.add(*[el.mul(0.01 * (k + 1), X(k % 2)) for k in range(100)])], 2),
    "phasor": (lambda: [el.phasor(440.0), el.phasor(el.add(300.0, el.mul(200.0, X())))], 1),
    "sphasor": (lambda: [el.syncphasor(440.0, el.train(37.0))], 0),
    "train_cycle": (lambda: [el.train(5.0), el.cycle(220.0)], 0),
    "blepsaw": (lambda: [el.blepsaw(440.0), el.blepsaw(el.add(1000.0, el.mul(800.0, X())))], 1),
    "blepsquare": (lambda: [el.blepsquare(311.0)], 0),
    "bleptriangle": (lambda: [el.bleptriangle(523.25)], 0),
    "rand": (lambda: [el.rand({"seed": 17}), el.noise({"seed": 99})], 0),
    "counter": (lambda: [el.counter(el.train(50.0))], 0),
    "accum": (lambda: [el.accum(el.abs(X()), el.train(20.0))], 1),
    "latch": (lambda: [el.latch(el.train(100.0), X())], 1),
    "maxhold": (lambda: [el.maxhold({"hold": 3.0}, el.abs(X()), el.train(9.0)), el.maxhold({}, X(), 0.0)], 1),
    "once": (lambda: [el.once({"arm": True}, el.train(30.0))], 0),
    "seq": (lambda: [el.seq({"seq": [1, 2, 3, 5.5], "hold": True}, el.train(200.0), 0),
                     el.seq({"seq": [0.5, 0.25], "loop": False}, el.train(150.0), el.train(7.0)),
                     el.seq({"seq": [3, 4, 5], "offset": 1, "hold": False}, el.train(90.0), el.train(11.0))], 0),
    "pole": (lambda: [el.pole(0.99, X()), el.pole(el.add(0.5, X(1)), X(0))], 2),
    "smooth_sm": (lambda: [el.sm(X()), el.smooth(0.95, el.train(3.0))], 1),
    "env": (lambda: [el.env(el.tau2pole(0.001), el.tau2pole(0.05), X())], 1),
    "biquad": (lambda: [el.biquad(0.2, 0.3, 0.2, -0.5, 0.2, X())], 1),
    "prewarp_mm1p": (lambda: [el.mm1p({"mode": "lowpass"}, el.prewarp(800.0), X()),
                              el.mm1p({"mode": "highpass"}, el.prewarp(el.add(1000, el.mul(900, X(1)))), X(0)),
                              el.mm1p({"mode": "allpass"}, 0.3, X())], 2),
    "svf_modes": (lambda: [el.lowpass(800, 1.0, X()), el.highpass(1200, 0.7, X()), el.bandpass(500, 4.0, X()),
                           el.notch(2000, 2.0, X()), el.allpass(900, 1.0, X())], 1),
    "svf_modulated": (lambda: [el.lowpass(el.add(1000, el.mul(900, el.cycle(3.0))), el.add(2.0, X(1)), X(0))], 2),
    "svfshelf": (lambda: [el.lowshelf(300, 0.8, 6.0, X()), el.highshelf(4000, 0.7, -4.5, X()), el.peak(1000, 2.0, 9.0, X())], 1),
    "z": (lambda: [el.z(X()), el.zero(0.5, 0.5, X()), el.dcblock(X())], 1),
    "sdelay": (lambda: [el.sdelay({"size": 10}, X()), el.sdelay({"size": 700}, X()), el.sdelay({"size": 0}, X())], 1),
    "delay_long": (lambda: [el.delay({"size": 4000}, 1500.5, 0.5, X()), el.delay({"size": 24000}, el.add(3000, el.mul(100, X(1))), 0.3, X(0))], 2),
    "delay_short": (lambda: [el.delay({"size": 100}, 10.25, 0.6, X()), el.delay({"size": 10}, 0.5, 0, X()),
                             el.delay({"size": 10}, 0, 0, X()), el.delay({"size": 2000}, el.add(300, el.mul(299, X(1))), -0.4, X(0))], 2),
    "taps": (lambda: [el.tapOut({"name": "fb"}, el.add(el.mul(0.5, el.tapIn({"name": "fb"})), X()))], 1),
    "time_metro": (lambda: [el.mul(1e-4, el.time()), el.metro({"interval": 3.0})], 0),
    "pink_noise": (lambda: [el.pinknoise({"seed": 5})], 0),
    "adsr": (lambda: [el.adsr(0.002, 0.01, 0.5, 0.02, el.train(10.0))], 0),
    "compress": (lambda: [el.compress(5, 50, -20, 4, X(), X())], 1),
    "shared_between_roots": (lambda: (lambda s: [el.mul(0.5, s), el.tanh(s), s])(el.cycle(330.0)), 0),
}


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
