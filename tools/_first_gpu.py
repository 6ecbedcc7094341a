import os as _os, sys as _sys; _R = _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))); _sys.path[:0] = [_R, _os.path.join(_R, 'tests')]
import sys, time, numpy as np
sys.path.insert(0, '.')
sys.path.insert(0, 'tests')
from elementary_amd import el, graphs
from elementary_amd.runtime import Runtime
from oracle import RefRuntime
from helpers import render_pair

def hip(sr, bs): return Runtime(sr, bs)
def ref(sr, bs): return RefRuntime(sr, bs)

cases = {
 "const": (lambda: [el.mul(0.5, 2.0)], 44100.0, 0),
 "c1": (graphs.c1_graph, 44100.0, 0),
 "in": (lambda: [el.mul(2.0, el.in_({"channel": 0}))], 44100.0, 1),
 "phasor": (lambda: [el.phasor(440.0)], 44100.0, 0),
 "blepsaw": (lambda: [el.blepsaw(440.0)], 44100.0, 0),
 "pole": (lambda: [el.pole(0.99, el.in_({"channel": 0}))], 44100.0, 1),
 "svf": (lambda: [el.lowpass(800, 1.0, el.in_({"channel": 0}))], 44100.0, 1),
 "tanh": (lambda: [el.tanh(el.in_({"channel": 0}))], 44100.0, 1),
 "voice": (lambda: [graphs.c2_voice(3)], 48000.0, 0),
 "c2_16": (lambda: graphs.c2_graph(16), 48000.0, 0),
}
for name, (fn, sr, nin) in cases.items():
    try:
        t = time.time()
        a, b = render_pair(hip, ref, fn, sample_rate=sr, blocks=12, n_in=nin)
        err = np.abs(a - b).max(axis=(1, 2))
        print(f"{name:10s} maxerr={err.max():.3e} max|ref|={np.abs(b).max():.3f} per-block={np.array2string(err, precision=1)} {time.time()-t:.2f}s", flush=True)
    except Exception as e:
        print(name, "FAILED", repr(e), flush=True)
