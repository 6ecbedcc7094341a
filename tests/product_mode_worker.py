"""Worker of tests/test_gpu_product_mode.py: runs in a process of its own with the PRODUCT defaults — ELEMHIP_SPECIALIZE=1
(specialised kernels compiled in the background, switched in mid-stream) and a COLD kernel cache directory, so that hiprtc
compiles on this machine. Every case renders launch sets against the reference engine from the first block (interpreter
kernels) until its specialised kernels have taken over and rendered three more sets. One JSON line on stdout."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]

import numpy as np
import torch

from elementary_amd import graphs
from elementary_amd.runtime import Runtime
import oracle
from helpers import lcg_noise

assert os.environ.get("ELEMHIP_SPECIALIZE") == "1" and os.environ.get("ELEMHIP_KCACHE")
# Background mode compiles a shape only when at least two islands of the plan have it (a one-off island renders through the
# interpreter kernel for good, plan.cpp), so the cases here are graphs of repeated islands: synth voices + their two mixers,
# independent render jobs (two shapes), feedback loops through taps (one block in flight).
# r05: a shape only ONE island has is compiled too — once its plan has rendered 64 blocks and lived 30 ms (Engine::promoteDeferredShapes):
# `one_off_chain` is a mono effects chain, a single island.
CASES = ["c2x16", "c4x8", "four_tap_loops", "one_off_chain"]


def make(name):
    n_in = 0
    if name == "c2x16":
        roots, sr = graphs.c2_graph(voices=16), graphs.C2_SAMPLE_RATE
    elif name == "c4x8":
        roots, sr = [graphs.c4_instance(k) for k in range(8)], graphs.C4_SAMPLE_RATE
    elif name == "one_off_chain":
        from elementary_amd import el
        x = el.in_({"channel": 0})
        roots, sr, n_in = [el.tanh(el.lowpass(700.0, 1.2, el.add(el.mul(0.3, el.cycle(330.0)), el.sdelay({"size": 300}, x))))], 44100.0, 1
    else:
        from elementary_amd import el     # four roots, a filtered feedback loop through a tap each: four islands of one shape

        def loop(k, x):
            fb = el.tapIn({"name": f"rv{k}"})
            body = el.lowpass(900.0 + 170.0 * k, 0.9, el.add(x, el.mul(0.7, el.sdelay({"size": 200 + 13 * k}, fb))))
            return el.tanh(el.tapOut({"name": f"rv{k}"}, body))
        roots, sr, n_in = [loop(k, el.in_({"channel": 0})) for k in range(4)], 44100.0, 1
    a = Runtime(sr, 512, device=0)
    c = oracle.RefRuntime(sr, 512) if oracle.have_ref() else oracle.PortRuntime(sr, 512)
    a.set_option("batch_blocks", 6)
    for rt in (a, c):
        assert rt.render(*roots)["result"] == 0          # (background mode: returns at once, the shapes are queued)
    return {"name": name, "a": a, "c": c, "n_in": n_in, "n_out": len(roots), "k": 0, "after": 0, "worst": 0.0, "scale": 1.0, "sets_interp": 0}


def step(e, nb=6):
    k0, n_in, n_out = e["k"], e["n_in"], e["n_out"]
    x = np.stack([lcg_noise(nb * 512, 17 + 97 * k0 + ch, 0.5) for ch in range(n_in)]) if n_in else None
    out = torch.zeros((nb, n_out, 512), dtype=torch.float32, device="cuda")
    before = e["a"].stats()["spec_launches"]
    if n_in:
        xin = torch.from_numpy(np.ascontiguousarray(x.reshape(n_in, nb, 512).transpose(1, 0, 2))).cuda()
        torch.cuda.synchronize()
        e["a"].process_blocks(nb, n_out, out_ptr=out.data_ptr(), in_ptr=xin.data_ptr(), num_inputs=n_in)
    else:
        torch.cuda.synchronize()
        e["a"].process_blocks(nb, n_out, out_ptr=out.data_ptr())
    got = out.cpu().numpy()
    ref = np.stack([e["c"].process(None if x is None else x[:, b * 512:(b + 1) * 512], n_out, 512) for b in range(nb)])
    e["scale"] = max(e["scale"], float(np.abs(ref).max()))
    e["worst"] = max(e["worst"], float(np.abs(got - ref).max()))
    e["k"] += nb
    if e["a"].stats()["spec_launches"] > before:
        e["after"] += 1
    else:
        e["sets_interp"] += 1


t0 = time.time()
engines = [make(n) for n in CASES]
deadline = t0 + float(os.environ.get("PRODUCT_MODE_BUDGET_S", "90"))
while time.time() < deadline and any(e["after"] < 3 for e in engines):
    for e in engines:
        if e["after"] < 3:
            step(e)
# the loop above ends once every engine has rendered three sets with SOME specialised kernel in place; the other shapes of its plan
# may still be in the compiler: wait for them (so that the count below does not depend on timing) and render one more set with
# every kernel loaded
def _all_compiled(e):
    return all(e["a"].spec_info(k)["state"] not in (0, 2) for k in range(e["a"].stats()["spec_shapes"]))   # (2: deferred, not queued yet)


while time.time() < deadline and not all(_all_compiled(e) for e in engines):
    time.sleep(0.1)
for e in engines:
    step(e)
rows = []
for e in engines:
    st = e["a"].stats()
    rows.append({"case": e["name"], "blocks": e["k"], "sets_through_interpreter": e["sets_interp"], "sets_through_specialised": e["after"],
                 "spec_shapes": st["spec_shapes"], "max_err": e["worst"], "scale": e["scale"]})
print(json.dumps({"seconds": time.time() - t0, "kcache": os.environ["ELEMHIP_KCACHE"], "compiled_here": len(os.listdir(os.environ["ELEMHIP_KCACHE"])), "rows": rows}))
