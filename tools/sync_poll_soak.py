"""Soak of the synchronous call's new ending (`sync_poll`: the host reads the output block as soon as the epilogue kernel's word
arrives in mapped host memory): examples/bench_cli renders the cli benchmark's graph for `calls` synchronous elemhip_process calls
and keeps EVERY block; every one of them is compared with the reference engine. A block read before its data had landed would
carry the previous block's samples (the graph is two sines: consecutive blocks differ everywhere).
Usage (GPU box): python tools/sync_poll_soak.py [calls=100000] [c1|c2]   (c2: the 256-voice synth, two launch levels, side streams)"""
import os as _os, sys as _sys; _R = _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))); _sys.path[:0] = [_R, _os.path.join(_R, 'tests')]
import json
import subprocess
import sys
import tempfile
import time

import numpy as np

import oracle
from elementary_amd import graphs
from elementary_amd.reconciler import Renderer, batch_to_json

calls = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
which = sys.argv[2] if len(sys.argv) > 2 else "c1"
sr, roots = (graphs.C2_SAMPLE_RATE, graphs.c2_graph(voices=256, channels=2)) if which == "c2" else (graphs.C1_SAMPLE_RATE, graphs.c1_graph())
sent = []
Renderer(lambda b: sent.append(b) or 0).render(*roots)
with tempfile.TemporaryDirectory() as d:
    bpath, dump = _os.path.join(d, "batch.json"), _os.path.join(d, "all.f32")
    open(bpath, "w").write(batch_to_json(sent[0]))
    t0 = time.time()
    res = subprocess.run([_os.path.join(_R, "examples", "bench_cli"), bpath, str(calls), str(sr), _os.path.join(d, "last.f32"), dump],
                         capture_output=True, text=True, timeout=1200, env=dict(_os.environ, ELEMHIP_SPECIALIZE="2"))
    assert res.returncode == 0, res.stderr[-400:]
    timing = json.loads([l for l in res.stderr.splitlines() if l.startswith("{")][-1])
    got = np.fromfile(dump, dtype=np.float32).reshape(calls + 1, 2, 512)
ref_rt = oracle.RefRuntime(sr, 512) if oracle.have_ref() else oracle.PortRuntime(sr, 512)
assert ref_rt.render(*roots)["result"] == 0
worst, bad, stale = 0.0, 0, 0
prev = None
for k in range(calls + 1):
    y = ref_rt.process(None, 2, 512)
    e = float(np.abs(got[k] - y).max())
    worst = max(worst, e)
    if e > 1e-6 * max(1.0, float(np.abs(y).max())):
        bad += 1
        if prev is not None and float(np.abs(got[k] - prev).max()) <= 1e-6:
            stale += 1
    prev = y
print(json.dumps({"graph": "C2 (256 voices)" if which == "c2" else "C1 (cli/Benchmark)", "calls": calls, "blocks_checked": calls + 1, "max_abs_err": worst, "blocks_over_1e-6": bad,
                  "of_them_equal_to_the_previous_block": stale, "timing_us": timing, "wall_s": round(time.time() - t0, 1)}))
assert bad == 0
