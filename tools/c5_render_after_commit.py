"""Synchronous elemhip_process latency as a function of the number of blocks since the last commit of the C5 mutation stream
(one voice replaced): which kernels render a live graph between two commits? One JSON line."""
import os as _os, sys as _sys; _R = _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))); _sys.path[:0] = [_R, _os.path.join(_R, 'benchmarks')]
import json
import time

import numpy as np
import torch  # noqa: F401

from elementary_amd import graphs
from elementary_amd.runtime import Runtime
import bench_configs as B

texts, _, _ = B._c5_batches(128, 40)
rt = Runtime(graphs.C2_SAMPLE_RATE, 512, device=0)
rt.set_option("specialize", 1)
assert rt.apply_instructions_json(texts[0]) == 0
for _ in range(300):
    rt.process(None, 2, 512)
quiet = []
for _ in range(300):
    t0 = time.perf_counter(); rt.process(None, 2, 512); quiet.append(1e6 * (time.perf_counter() - t0))
pos = [[] for _ in range(40)]
info = []
for k in range(1, len(texts)):
    assert rt.apply_instructions_json(texts[k]) == 0
    for j in range(40):
        t0 = time.perf_counter(); rt.process(None, 2, 512); pos[j].append(1e6 * (time.perf_counter() - t0))
    if k % 16 == 0:
        rt.gc()
    st, pl = rt.stats(), rt.describe_plan()
    info.append((st["spec_shapes"], st["spec_islands"], pl["num_islands"], pl["level_sizes"]))
print(json.dumps({"quiet_us_p50": float(np.median(quiet)), "us_p50_by_blocks_since_commit": [round(float(np.median(p)), 1) for p in pos],
                  "spec_launches": rt.stats()["spec_launches"], "plans": info[-3:]}))
