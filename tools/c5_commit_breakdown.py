"""Where a commit of the C5 mutation stream goes on the GPU box: per instruction kind (ELEMHIP_APPLY_TIMING, engine.cpp apply),
per plan-build phase (describe_plan build_us), the call as Python sees it, the first block after it.
Usage: python tools/c5_commit_breakdown.py [batches]   (one JSON line)"""
import os as _os, sys as _sys; _R = _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))); _sys.path[:0] = [_R, _os.path.join(_R, 'benchmarks')]
import json
import os
import re
import subprocess
import sys
import time

import numpy as np


def child(n):
    import torch
    from elementary_amd import graphs
    from elementary_amd.runtime import Runtime
    import bench_configs as B
    texts, _, _ = B._c5_batches(128, n)
    rt = Runtime(graphs.C2_SAMPLE_RATE, 512, device=0)
    rt.set_option("specialize", 1)
    out = torch.empty((64, 2, 512), dtype=torch.float32, device="cuda")
    assert rt.apply_instructions_json(texts[0]) == 0
    for _ in range(8):
        rt.process_blocks(64, 2, out_ptr=out.data_ptr())
    rows = []
    for k in range(1, len(texts)):
        t0 = time.perf_counter()
        assert rt.apply_instructions_json(texts[k]) == 0
        t1 = time.perf_counter()
        rt.process_blocks(1, 2, out_ptr=out.data_ptr())
        t2 = time.perf_counter()
        rows.append({"call_us": 1e6 * (t1 - t0), "first_block_us": 1e6 * (t2 - t1), **rt.describe_plan()["build_us"]})
        rt.process_blocks(64, 2, out_ptr=out.data_ptr())
        time.sleep(0.02)
        if k % 16 == 0:
            rt.gc()
    print("ROWS " + json.dumps(rows), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--child":
        child(int(sys.argv[2])); sys.exit(0)
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 80
    env = dict(os.environ, ELEMHIP_APPLY_TIMING="1")
    r = subprocess.run([sys.executable, __file__, "--child", str(n)], env=env, capture_output=True, text=True, timeout=500)
    rows = json.loads([l for l in r.stdout.splitlines() if l.startswith("ROWS ")][0][5:])
    kinds = [[float(v) for v in re.findall(r"[\d.]+(?= )|[\d.]+(?= us)", l.split("apply:")[1])] for l in r.stderr.splitlines() if "apply:" in l][1:]
    med = lambda a: float(np.median(np.asarray(a)))     # noqa: E731
    p99 = lambda a: float(np.percentile(np.asarray(a), 99))     # noqa: E731
    out = {"batches": len(rows)}
    for key in rows[0]:
        out[key + "_p50"] = med([r_[key] for r_ in rows[4:]]); out[key + "_p99"] = p99([r_[key] for r_ in rows[4:]])
    names = ["create", "delete", "append", "set", "activate", "commit"]
    for i, nm in enumerate(names):
        out["apply_" + nm + "_us_p50"] = med([k[i] for k in kinds[4:]])
    slow = sorted(range(4, len(rows)), key=lambda i: -rows[i]["call_us"])[:6]
    out["slowest_calls"] = [dict(batch=i + 1, after_gc=((i + 1) % 16 == 1), **{k: round(v, 1) for k, v in rows[i].items()}) for i in slow]
    print(json.dumps(out))
