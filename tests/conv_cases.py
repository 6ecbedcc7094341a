"""`convolve` parity scenarios (tests/golden/convolve_scenarios.json) restated for the Python-side
engines; the same file drives tests/golden/make_convolve_golden.js, which recorded
tests/golden/convolve_wasm.f32 from the reference's prebuilt wasm engine (wasm/Convolve.h:23-92)."""
import json
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SPEC = json.load(open(os.path.join(HERE, "golden", "convolve_scenarios.json")))
SCENARIOS = {s["name"]: s for s in SPEC["scenarios"]}
_MANIFEST = json.load(open(os.path.join(HERE, "golden", "convolve_wasm.json")))
_BLOB = np.fromfile(os.path.join(HERE, "golden", "convolve_wasm.f32"), dtype="<f4")
SR, BLOCK = 48000.0, 512
BATCH = [[0, 1, "root"], [0, 2, "convolve"], [0, 3, "in"], [3, 3, "channel", 0], [3, 1, "channel", 0],
         [2, 2, 3, 0], [2, 1, 2, 0], [4, [1]], [5]]


def lcg_stream(seed, n):
    """s = 1664525 s + 1013904223 mod 2^32, x = s / 2^31 - 1 (float64)."""
    out = np.empty(n, dtype=np.float64)
    s = seed & 0xFFFFFFFF
    for i in range(n):
        s = (1664525 * s + 1013904223) & 0xFFFFFFFF
        out[i] = s / 2147483648.0 - 1.0
    return out


def make_ir(spec):
    n = spec["len"]
    noise = lcg_stream(spec["seed"], n)
    ir = np.empty(n, dtype=np.float64)
    d, r = 1.0, float(spec["r"])
    for k in range(n):                      # sequential products: bit-identical to the JS generator
        ir[k] = noise[k] * d
        d = d * r
    ir[0] = 1.0
    if "tiny_from" in spec:
        ir[spec["tiny_from"]:] = ir[spec["tiny_from"]:] * 1e-7
    ss = 0.0
    for v in ir:
        ss = ss + v * v
    return (ir / np.sqrt(ss)).astype(np.float32)


def golden(name):
    m = _MANIFEST[name]
    return _BLOB[m["offset"]:m["offset"] + m["count"]]


def total_frames(sc):
    return sum(c[1] * c[2] for c in sc["calls"] if c[0] == "run")


def scenario_input(sc):
    return (lcg_stream(sc["input_seed"], total_frames(sc)) * 0.25).astype(np.float32)


def run_scenario(make, name):
    """Drive an engine (apply_instructions / add_shared_resource / process) through a scenario."""
    sc = SCENARIOS[name]
    rt = make(SR, BLOCK)
    for spec in sc["irs"]:
        assert rt.add_shared_resource(spec["name"], make_ir(spec))
    assert rt.apply_instructions(BATCH) == 0
    x = scenario_input(sc)
    pos, out = 0, []
    for call in sc["calls"]:
        if call[0] == "path":
            assert rt.apply_instructions([[3, 2, "path", call[1]], [5]]) == 0
        else:
            n = call[1]
            for _ in range(call[2]):
                out.append(rt.process(x[None, pos:pos + n], 1, n)[0].copy())
                pos += n
    return np.concatenate(out)


def exact_model(name):
    """float64 linear convolution + the root's 20 ms fade-in: what every convolver approximates."""
    sc = SCENARIOS[name]
    irs = {s["name"]: make_ir(s).astype(np.float64) for s in sc["irs"]}
    x = scenario_input(sc).astype(np.float64)
    y = np.zeros_like(x)
    pos, cur, start = 0, None, 0
    spans = []
    for call in sc["calls"]:
        if call[0] == "path":
            if cur is not None:
                spans.append((cur, start, pos))
            cur, start = call[1], pos
        else:
            pos += call[1] * call[2]
    if cur is not None:
        spans.append((cur, start, pos))
    for irname, a, b in spans:                # a new IR starts from an empty input history (Convolve.h:47-51)
        h = irs[irname].copy()
        h[np.abs(h) < 1e-6] *= 1.0            # (tail trimming only drops |h| < 1e-6: below the tolerance)
        from scipy.signal import fftconvolve
        y[a:b] = fftconvolve(x[a:b], h)[:b - a]
    step = np.float32(1.0 / (SR * 20.0 / 1000.0))
    gain = np.minimum(1.0, np.arange(len(y)) * np.float64(step))
    return y * gain
