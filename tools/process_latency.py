"""Block-at-a-time latency of elemhip_process (the cli/Benchmark.cpp:70-101 protocol: one synchronous call per block) on the C2
graph, with the per-level device times of the launch profile beside the wall clock.
Usage: python tools/process_latency.py [voices] [blocks]"""
import os as _os, sys as _sys; _R = _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))); _sys.path[:0] = [_R, _os.path.join(_R, 'tests')]
import json
import sys
import time

import numpy as np
import torch  # noqa: F401

from elementary_amd import graphs
from elementary_amd.runtime import Runtime

which = sys.argv[1] if len(sys.argv) > 1 else "256"     # a voice count of the C2 graph, "c1" (cli/Benchmark graph) or "floor" (a constant per channel)
voices = int(which) if which.isdigit() else which
blocks = int(sys.argv[2]) if len(sys.argv) > 2 else 400
for spec_blocks, direct, graph in ((0, 1, 0), (1, 0, 0), (1, 1, 0), (1, 1, 1)):
    rt = Runtime(graphs.C1_SAMPLE_RATE if which == "c1" else graphs.C2_SAMPLE_RATE, 512, device=0)
    rt.set_option("specialize", 2)
    rt.set_option("spec_blocks", spec_blocks)
    rt.set_option("host_out_direct", direct)
    rt.set_option("spec_block_graph", graph)
    from elementary_amd import el
    roots = graphs.c1_graph() if which == "c1" else [el.const({"value": 0.25}), el.const({"value": 0.5})] if which == "floor" else graphs.c2_graph(voices=voices)
    assert rt.render(*roots)["result"] == 0
    for _ in range(60):
        rt.process(None, 2, 512)
    ts = []
    for _ in range(blocks):
        t0 = time.perf_counter()
        rt.process(None, 2, 512)
        ts.append(time.perf_counter() - t0)
    ts = np.array(ts) * 1e6
    rt.set_option("profile_launches", 1)
    for _ in range(100):
        rt.process(None, 2, 512)
    prof = rt.launch_profile()
    rt.set_option("profile_launches", 0)
    print(json.dumps({"voices": voices, "spec_blocks": spec_blocks, "host_out_direct": direct, "spec_block_graph": graph, "us_mean": float(ts.mean()), "us_p50": float(np.percentile(ts, 50)),
                      "us_p99": float(np.percentile(ts, 99)), "device_level_us": [1e3 * x / max(1, prof["blocks"]) for x in prof["level_ms"]],
                      "device_epilogue_us": 1e3 * prof["epilogue_ms"] / max(1, prof["blocks"]), "profiled_blocks": prof["blocks"]}), flush=True)
