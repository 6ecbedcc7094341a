"""Diagnostic: node cases through the single-block specialised path (elemhip_process, specialize = 2) vs the interpreter kernels
(spec_blocks = 0) vs the reference engine: per case, the first block and output that differ. Usage: python tools/diag_single_block_spec.py [case ...]"""
import os as _os, sys as _sys; _R = _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))); _sys.path[:0] = [_R, _os.path.join(_R, 'tests')]
import json
import sys

import numpy as np
import torch  # noqa: F401

from elementary_amd.runtime import Runtime
from cases import NODE_CASES, REF_ONLY, node_case_resources
from helpers import lcg_noise
import oracle

names = sys.argv[1:] or sorted(NODE_CASES)
for name in names:
    roots_fn, n_in = NODE_CASES[name]
    a, b = Runtime(44100.0, 512, device=0), Runtime(44100.0, 512, device=0)
    c = oracle.RefRuntime(44100.0, 512) if oracle.have_ref() else (None if name in REF_ONLY else oracle.PortRuntime(44100.0, 512))
    a.set_option("specialize", 2); b.set_option("specialize", 0)
    rts = [r for r in (a, b, c) if r is not None]
    for rt in rts:
        for rname, data in node_case_resources().items():
            assert rt.add_shared_resource(rname, data)
    roots = roots_fn()
    n_out = len(roots)
    for rt in rts:
        assert rt.render(*roots)["result"] == 0
    row = {"case": name}
    for k in range(12):
        x = np.stack([lcg_noise(512, 1 + ch + 97 * k, 0.5) for ch in range(max(n_in, 1))])
        ys = [rt.process(x if n_in else None, n_out, 512) for rt in rts]
        scale = max(1.0, float(np.abs(ys[-1]).max()))
        d_ai = np.abs(ys[0] - ys[1]).max(axis=1)
        d_ar = np.abs(ys[0] - ys[-1]).max(axis=1)
        if (d_ai > 1e-6 * scale).any() or (d_ar > 1e-6 * scale).any():
            ch = int(np.argmax(np.maximum(d_ai, d_ar)))
            fr = np.nonzero(np.abs(ys[0][ch] - ys[-1][ch]) > 1e-6 * scale)[0]
            row.update(first_bad_block=k, channel=ch, spec_vs_interp=float(d_ai.max()), spec_vs_ref=float(d_ar.max()), interp_vs_ref=float(np.abs(ys[1] - ys[-1]).max()),
                       frames=[int(fr.min()), int(fr.max()), int(fr.size)] if fr.size else [], spec=[float(v) for v in ys[0][ch][fr[:4]]] if fr.size else [], ref=[float(v) for v in ys[-1][ch][fr[:4]]] if fr.size else [])
            break
    info = a.describe_plan()
    row.update(spec_fade_blocks=info["plan_spec_fade_blocks"], spec_launches=a.stats()["spec_launches"])
    if "first_bad_block" in row or len(names) < 5:
        print(json.dumps(row), flush=True)
print("done")
