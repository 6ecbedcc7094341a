"""Fill the on-disk kernel cache (elementary_amd/kcache/) with the specialised island kernels of the graphs the test
suite and the benchmarks render, by building their plans on a dry engine (no GPU needed: hiprtc cross-compiles for
gfx950). The cache travels with the tree; a miss only costs the compile at first use.
Usage: python tools/warm_kcache.py [bench|tests|all] [i/n]"""
import os as _os, sys as _sys; _R = _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))); _sys.path[:0] = [_R, _os.path.join(_R, 'tests'), _os.path.join(_R, 'tools')]
import sys
import time

import torch  # noqa: F401  (first: the GPU-side processes import torch before the engine, and hiprtc binds accordingly)

from elementary_amd import el, graphs
from elementary_amd.runtime import Runtime


def warm(sr, bs, roots, resources=None, batch=None, copies=None):
    rt = Runtime(sr, bs, device=-1)
    rt.set_option("specialize", 2)
    if copies is not None:
        rt.set_option("pipeline_copies", copies)
    for name, data in (resources or {}).items():
        rt.add_shared_resource(name, data)
    res = rt.render(*roots)
    assert res["result"] == 0, res["result"]
    st = rt.stats()
    k, bad = 0, 0
    while k < st["spec_shapes"]:
        info = rt.spec_info(k)
        if info["state"] != 1:
            bad += 1
            print(info["log"][:1500])
        k += 1
    return st["spec_shapes"], bad, st["last_jit_wait_ms"]


def main():
    what = sys.argv[1] if len(sys.argv) > 1 else "all"
    t0 = time.time()
    shapes = bad = 0
    jobs = []
    if what in ("bench", "all"):
        # the graphs bench.py renders, at full size: the 128-input mixers of the 256-voice graph are a shape of their own
        jobs.append(("c2", graphs.C2_SAMPLE_RATE, 512, graphs.c2_graph(), None, None))
        jobs.append(("c2x8", graphs.C2_SAMPLE_RATE, 512, graphs.c2_graph(voices=8), None, None))       # __graft_entry__.smoke()
        jobs.append(("c4", graphs.C4_SAMPLE_RATE, 512, [graphs.c4_instance(k) for k in range(128)], None, None))
        jobs.append(("c1", graphs.C1_SAMPLE_RATE, 512, graphs.c1_graph(), None, None))
    if what in ("tests", "all"):
        from cases import NODE_CASES, node_case_resources
        from test_gpu_fuzz import random_graph
        for name in sorted(NODE_CASES):
            jobs.append((name, 44100.0, 512, NODE_CASES[name][0](), node_case_resources(), None))
        for seed in range(0, 48, 3):
            n_out = min(3, 1 + seed % 5)
            jobs.append((f"fuzz{seed}", 48000.0, 512, random_graph(seed, n_nodes=24 + 22 * (seed % 4), n_roots=1 + seed % 5)[:n_out], None, None))
        jobs.append(("c2x16", 48000.0, 512, graphs.c2_graph(voices=16), None, None))
        from cases import every_stateful_roots
        for bs in (64, 192, 256):                      # tests/test_gpu_spec.py::test_spec_other_block_sizes
            jobs.append((f"c2x16_bs{bs}", 48000.0, bs, graphs.c2_graph(voices=16), None, None))
            jobs.append((f"stateful_bs{bs}", 48000.0, bs, every_stateful_roots(), None, None))
        import test_gpu_taps, tap_soak                  # tap islands (one block in flight) and the 8-loop soak graph
        for name, (fn, _n_in) in sorted(tap_soak.GRAPHS.items()):
            jobs.append((f"taps_{name}", 44100.0, 512, fn(), None, None))
        for name, case in sorted(test_gpu_taps.CASES.items()):
            if not case[2]: jobs.append((f"taps_{name}", 44100.0, 512, case[0](), None, None))
        jobs.append(("tap_loop_bench", 48000.0, 512, tap_soak._bench_graph(), None, None))
        import test_gpu_spec                              # tests/test_gpu_spec.py::test_fused_epilogue_of_a_single_block_call
        for name, (sr, mk, _n_in) in sorted(test_gpu_spec._fuse_graphs().items()):
            jobs.append((f"fuse_{name}", sr, 512, mk(), None, None))
            jobs.append((f"fuse_{name}_swapped", sr, 512, mk()[::-1], None, None))
        jobs.append(("phasors_only", 44100.0, 512, test_gpu_spec._phasors_only_roots(), node_case_resources(), None))
        for copies in (1, 3, 6):
            jobs.append((f"stateful_d{copies}", 48000.0, 512, every_stateful_roots(), None, copies))
    if len(sys.argv) > 2:      # "i/n": this process takes every n-th job (several processes warm the cache in parallel)
        i, n = (int(v) for v in sys.argv[2].split("/"))
        jobs = jobs[i::n]
    for name, sr, bs, roots, res, copies in jobs:
        s, b, ms = warm(sr, bs, roots, res, copies=copies)
        shapes += s; bad += b
        print(f"{name:28s} shapes {s}  failed {b}  waited {ms:8.0f} ms", flush=True)
    print(f"{shapes} shapes, {bad} failed, {time.time() - t0:.1f} s")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
