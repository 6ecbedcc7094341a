"""Graph cases shared by the CPU (oracle) and GPU (HIP engine) parity suites."""
import math  # noqa: F401

from elementary_amd import el, graphs  # noqa: F401

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
    # SURVEY 8(f) rank 2
    "table": (lambda: [el.table({"path": "/t/ramp"}, el.add(0.5, X())), el.table({"path": "/t/five"}, el.phasor(3.0)),
                       el.table({"path": "/t/one"}, X())], 1),
    "seq2": (lambda: [el.seq2({"seq": [1, 2, 3, 5.5], "hold": True}, el.train(200.0), 0),
                      el.seq2({"seq": [0.5, 0.25, 4], "loop": False}, el.train(150.0), el.train(7.0)),
                      el.seq2({"seq": [3, 4, 5], "offset": 1, "hold": False, "loop": True}, el.train(90.0), el.train(11.0)),
                      el.seq2({"seq": [7, 8], "loop": False, "hold": True}, el.train(400.0), el.train(5.0))], 0),
    "sparseq2": (lambda: [el.sparseq2({"seq": [{"time": 100, "value": 1}, {"time": 700, "value": 2.5}, {"time": 2000, "value": -1},
                                               {"time": 100, "value": 9}]}, el.time()),
                          el.sparseq2({"interpolate": 1, "seq": [{"time": 0, "value": 0}, {"time": 1000, "value": 1}, {"time": 3000, "value": -2}]},
                                      el.add(500, el.mod(el.time(), 2800))),
                          el.sparseq2({"interpolate": 1, "seq": [{"time": 0.25, "value": 1}, {"time": 0.5, "value": 3}]}, el.add(0.4, el.mul(0.8, X())))], 1),
    "sparseq": (lambda: [el.sparseq({"seq": [{"value": 1, "tickTime": 0}, {"value": 2, "tickTime": 2}, {"value": -3, "tickTime": 5}, {"value": 4, "tickTime": 9},
                                             {"value": 7, "tickTime": 2}], "loop": [2, 7]}, el.train(210.0), el.train(3.0)),
                         el.sparseq({"seq": [{"value": 0.5, "tickTime": 1}, {"value": 2.5, "tickTime": 4}, {"value": -1, "tickTime": 12}],
                                     "interpolate": 1, "tickInterval": 0.004, "offset": 1}, el.train(250.0), 0),
                         el.sparseq({"seq": [{"value": 3, "tickTime": 3}, {"value": 6, "tickTime": 40}], "interpolate": 1}, el.train(900.0), el.train(11.0)),
                         el.sparseq({"seq": [{"value": 1, "tickTime": 0}], "loop": False}, X(), 0)], 1),
    "capture": (lambda: [el.capture({"name": "c"}, el.train(15.0), X()), el.capture({}, el.ge(X(1), 0.0), el.mul(2.0, X(0)))], 2),
    "sample": (lambda: [el.sample({"path": "/t/ramp"}, el.train(90.0), 1.0),
                        el.sample({"path": "/t/ramp", "mode": "gate"}, el.train(40.0), el.add(1.5, X())),
                        el.sample({"path": "/t/ramp", "mode": "loop", "startOffset": 20, "stopOffset": 50}, el.train(5.0), 0.37),
                        el.sample({"path": "/t/five", "mode": "loop"}, el.train(300.0), 0.25),
                        el.sample({"path": "/t/ramp", "mode": "trigger", "startOffset": 10}, el.train(700.0), 2.0)], 1),
    # multi-output nodes, one root per output channel (the port oracle has no mc nodes: reference only)
    "mc": (lambda: el.mc.table({"path": "/t/stereo", "channels": 2}, el.add(0.5, X()))
                   + el.mc.sample({"path": "/t/stereo", "channels": 2, "mode": "loop", "playbackRate": 0.75, "startOffset": 7}, el.train(9.0))
                   + el.mc.sample({"path": "/t/stereo", "channels": 2, "mode": "gate"}, el.train(31.0))
                   + el.mc.sampleseq({"path": "/t/stereo", "channels": 2, "duration": 300,
                                      "seq": [{"time": 0, "value": 1}, {"time": 900, "value": 0}, {"time": 2000, "value": 1}, {"time": 5000, "value": 0}]},
                                     el.counter(1)), 1),
    # builtins/mc/Capture.h: pass-through of the inputs behind the gate (three capture channels, two of them consumed; one consumed)
    "mc_capture": (lambda: el.mc.capture({"name": "take", "channels": 2}, el.train(23.0), X(0), el.mul(0.5, X(1)), el.cycle(300.0))
                           + el.mc.capture({"channels": 1}, el.le(X(1), 0.1), el.mul(2.0, X(0))), 2),
}

# shared resources the cases above load (name -> channel-0 samples)
REF_ONLY = {"mc", "mc_capture"}      # cases the plain-C port oracle cannot render (checked against oracle/_ref only)


def node_case_resources():
    import numpy as np
    return {"/t/ramp": (np.arange(300, dtype=np.float32) / 300.0 + 0.25).astype(np.float32),
            "/t/five": np.asarray([1, 2, 3, 4, 5], np.float32), "/t/one": np.asarray([0.75], np.float32),
            "/t/stereo": np.stack([np.arange(300, dtype=np.float32) / 300.0 + 0.25, np.cos(np.arange(300, dtype=np.float32) * 0.05)]).astype(np.float32)}




def sampleseq_scenario(make, block=32, sr=44100.0):
    """sampleseq.test.js:5-76 ("sampleseq basics") extended: the time input is a ref'd const the
    host steps; covers onset/offset fades, a backwards jump, a seq swap, a duration change, a new
    sample buffer and running past the end of the sample. Returns [blocks, 1, block]."""
    import numpy as np
    rt = make(sr, block)
    ramp = (np.arange(300, dtype=np.float32) / 300.0 + 0.25).astype(np.float32)
    assert rt.add_shared_resource("/v/ones", np.ones(128, np.float32))
    assert rt.add_shared_resource("/v/ramp", ramp)
    t_node, set_time = rt.renderer.create_ref("const", {"value": 0}, [])
    seq_node, set_seq = rt.renderer.create_ref("sampleseq", {
        "duration": 128, "path": "/v/ones",
        "seq": [{"time": 0, "value": 0}, {"time": 128, "value": 1}, {"time": 256, "value": 0}, {"time": 512, "value": 1}],
    }, [t_node])
    assert rt.render(seq_node)["result"] == 0
    ys = []

    def run(times):
        for t in times:
            if t is not None:
                assert set_time({"value": t}) == 0
            ys.append(rt.process(None, 1, block))

    run([None] * 4)                                   # before the first onset; root fade-in settles
    run([129, 129 + block, 129 + 2 * block])         # onset: fade in 0, .02, .04 ...
    run([260, 260 + block])                           # offset: fade out
    run([64, 520, 520 + block, 520 + 2 * block, 520 + 3 * block, 520 + 4 * block])   # backwards jump, 2nd onset, run off the end
    assert set_seq({"seq": [{"time": 10, "value": 1}, {"time": 700, "value": 0}, {"time": 10, "value": 0}]}) == 0
    run([40, 40 + block, 300])                        # new seq (duplicate time keeps the first entry); misaligned jump
    assert set_seq({"duration": 300, "path": "/v/ramp"}) == 0
    run([100, 100 + block, 100 + 2 * block, 720, 0])
    return np.stack(ys)


def every_stateful_roots():
    """One graph with every stateful node type, host input 0 and the time-dependent nodes (pipelining tests)."""
    X = el.in_({"channel": 0})
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
