"""Per-op latency probe: 64 independent single-op islands, HIP-event kernel time."""
import os as _os, sys as _sys; _R = _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))); _sys.path[:0] = [_R, _os.path.join(_R, 'tests')]

import sys
sys.path.insert(0, '.')
from elementary_amd import el, graphs
from elementary_amd.runtime import Runtime

def probe(name, fn, nroots=64, sr=48000.0):
    rt = Runtime(sr, 512)
    assert rt.render(*[fn(k) for k in range(nroots)])["result"] == 0
    rt.time_launches(nroots, 20)
    lv = rt.time_launches(nroots, 200)
    p = rt.describe_plan()
    print(f"{name:14s} levels={len(lv)-1} us={[round(1e3*x,1) for x in lv]} islands={p['num_islands']} tasks/island={p['islands'][0]['tasks']} stages={p['islands'][0]['stages']} lds={p['islands'][0]['lds_bytes']}", flush=True)

K = lambda k: el.const({"key": f"k{k}", "value": 100.0 + k})
probe("const", lambda k: K(k))
probe("mul1", lambda k: el.mul(K(k), 0.5))
def chain(k, n):
    x = K(k)
    for i in range(n): x = el.mul(x, 1.0001 + i * 1e-6)
    return x
probe("mul4", lambda k: chain(k, 4))
probe("mul16", lambda k: chain(k, 16))
probe("tanh", lambda k: el.tanh(el.phasor(K(k))))
probe("phasor", lambda k: el.phasor(K(k)))
probe("blepsaw", lambda k: el.blepsaw(K(k)))
probe("blepsaw2", lambda k: el.add(el.blepsaw(K(k)), el.blepsaw(el.const({"key": f"j{k}", "value": 200.0 + k}))))
probe("pole", lambda k: el.pole(0.999, el.phasor(K(k))))
probe("svf", lambda k: el.svf({"mode": "lowpass"}, 800.0, 2.0, el.phasor(K(k))))
probe("svf_hp", lambda k: el.svf({"mode": "highpass"}, 800.0, 2.0, el.phasor(K(k))))
probe("biquad", lambda k: el.biquad(0.2, 0.3, 0.2, -0.5, 0.2, el.phasor(K(k))))
probe("voice", lambda k: graphs.c2_voice(k))
probe("add128", lambda k: el.add(*[el.phasor(el.const({"key": f"a{k}_{j}", "value": 50.0 + j})) for j in range(128)]), nroots=2)
