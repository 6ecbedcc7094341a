"""Shape churn at the product default (VERDICT r04 "next" #2): `specialize` = 1 (kernels compiled in the background) under a
stream in which EVERY commit brings an island shape nobody has seen — what a live-coding session does to the engine. The
reference has no compile step (runtime/elem/Runtime.h:170-218, 277-285): every commit must render correctly from its first
block whatever the compiler is doing, and nothing may grow without bound: kernel-cache entries, loaded code objects
(hipModuleUnload once no plan references a shape), host memory, device memory. Also: a STATIC one-off graph leaves the
interpreter kernel (r04: never)."""
import os
import sys
import time

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "benchmarks"))
pytestmark = pytest.mark.gpu
TOL = 1e-6


def test_shape_churn_soak_2000_commits(gpu_required):
    import psutil
    import torch
    import oracle
    import driver_configs as dc
    from elementary_amd import graphs
    from elementary_amd.runtime import Runtime
    voices, commits, after, cap = 32, 2000, 3, 48
    texts, creates, sizes = dc._c5_batches(voices, commits, churn=True)
    hip = Runtime(graphs.C2_SAMPLE_RATE, 512, device=0)
    hip.set_option("specialize", 1)
    hip.set_option("jit_cache_entries", cap)
    ref = oracle.RefRuntime(graphs.C2_SAMPLE_RATE, 512) if oracle.have_ref() else oracle.PortRuntime(graphs.C2_SAMPLE_RATE, 512)
    for rt in (hip, ref):
        assert rt.apply_instructions_json(texts[0]) == 0
    proc = psutil.Process()
    worst, scale, blocks = 0.0, 1.0, 0
    marks = {}
    t0 = time.time()
    for k in range(1, commits + 1):
        assert hip.apply_instructions_json(texts[k]) == 0
        assert ref.apply_instructions_json(texts[k]) == 0
        for _ in range(after):
            a, b = hip.process(None, 2, 512), ref.process(None, 2, 512)
            scale = max(scale, float(np.abs(b).max()))
            worst = max(worst, float(np.abs(a - b).max()))
            blocks += 1
        if k % 16 == 0:
            assert sorted(hip.gc()) == sorted(ref.gc())
        if k in (400, commits):
            torch.cuda.synchronize()
            jit = hip.describe_plan()["jit"]
            marks[k] = {"rss": proc.memory_info().rss, "dev_free": torch.cuda.mem_get_info()[0], "jit": jit}
    plan = hip.describe_plan()
    jit = plan["jit"]
    print(f"{commits} commits, {blocks} blocks in {time.time() - t0:.1f} s; worst err {worst:.2e} (scale {scale:.2f}); interpreter share of island-blocks "
          f"{plan['interp_block_fraction']:.3f}; jit {jit}; rss {marks[400]['rss'] >> 20} -> {marks[commits]['rss'] >> 20} MB; "
          f"device free {marks[400]['dev_free'] >> 20} -> {marks[commits]['dev_free'] >> 20} MB")
    assert np.isfinite(worst) and worst <= TOL * max(1.0, scale)
    live = plan["shapes"]["total"]
    assert jit["entries"] <= cap + live + 8, jit                       # the table is capped (entries of live plans may sit above the cap)
    assert jit["modules_loaded"] <= jit["entries"], jit                 # evicted shapes were unloaded
    assert jit["evictions"] > 0 and jit["text_bytes_held"] <= (cap + live + 8) * 256 * 1024, jit
    assert marks[commits]["rss"] - marks[400]["rss"] < 256 << 20, marks               # host memory: flat after warm-up
    assert marks[400]["dev_free"] - marks[commits]["dev_free"] < 512 << 20, marks     # device memory: program heaps / tables recycled


def test_a_static_one_off_graph_leaves_the_interpreter(gpu_required, tmp_path):
    """The reference's `cli` use case: ONE patch, rendered for a long time. Its islands are all one of a kind, so background mode
    used to leave them on the interpreter kernel for good (r04: `lonely` shapes never compiled). Now the shapes are queued once the
    plan has rendered `spec_lonely_blocks` blocks and lived `spec_lonely_ms`; the stream switches kernels without a sample moving."""
    import subprocess
    import json
    code = r'''
import json, os, sys, time
sys.path[:0] = [%r, %r]
import numpy as np, torch
from elementary_amd import el
from elementary_amd.runtime import Runtime
import oracle
rt = Runtime(44100.0, 512, device=0)
ref = oracle.RefRuntime(44100.0, 512) if oracle.have_ref() else oracle.PortRuntime(44100.0, 512)
x = el.in_({"channel": 0})
patch = [el.tanh(el.lowpass(800.0, 1.0, el.add(el.mul(0.3, el.cycle(440.0)), el.delay({"size": 4000}, 1200.0, 0.4, x))))]
for r in (rt, ref):
    assert r.render(*patch)["result"] == 0
first = rt.describe_plan()["shapes"]
worst, k, t0, switched_at = 0.0, 0, time.time(), None
rng = np.random.default_rng(1)
while time.time() - t0 < 60.0:
    xin = (rng.random((1, 512), dtype=np.float32) - 0.5).astype(np.float32)
    a, b = rt.process(xin, 1, 512), ref.process(xin, 1, 512)
    worst = max(worst, float(np.abs(a - b).max())); k += 1
    if switched_at is None and rt.stats()["spec_launches"] > 0:
        switched_at = k
    if switched_at is not None and k >= switched_at + 200:
        break
plan = rt.describe_plan()
print(json.dumps({"first": first, "last": plan["shapes"], "switched_at_block": switched_at, "blocks": k, "worst": worst, "seconds": time.time() - t0,
                  "jit": plan["jit"], "interp_block_fraction": plan["interp_block_fraction"]}))
''' % (ROOT, os.path.join(ROOT, "tests"))
    env = dict(os.environ, ELEMHIP_SPECIALIZE="1", ELEMHIP_KCACHE=str(tmp_path))
    res = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, cwd=ROOT, env=env)
    assert res.returncode == 0, (res.stdout[-1500:], res.stderr[-3000:])
    out = json.loads([ln for ln in res.stdout.splitlines() if ln.startswith("{")][-1])
    print(out)
    assert out["first"]["deferred"] >= 1 and out["first"]["ready"] == 0, out          # committed without waiting, nothing queued yet
    assert out["switched_at_block"] is not None and out["switched_at_block"] >= 64, out   # ... promoted after 64 blocks, compiled, switched in
    assert out["last"]["ready"] == out["last"]["total"], out
    assert out["worst"] <= TOL, out
    assert out["jit"]["compiles"] >= 1 and out["jit"]["promoted"] >= 1, out
