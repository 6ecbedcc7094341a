"""Host side of the HIP engine without a GPU: the C-ABI library loads and exports every symbol
include/elemhip.h declares, and a "dry" handle (deviceOrdinal -1: instruction decode, graph
mutation, plan build and gc, no rendering) behaves like the reference for return codes and gc."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import oracle
from elementary_amd import el, graphs
from elementary_amd.runtime import Runtime, load_library

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_abi_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "elemhip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    names = sorted(set(re.findall(r"\b(elemhip_[a-z_]+)\s*\(", hdr)))
    assert len(names) >= 20
    lib = load_library()
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing


def test_no_gpu_means_no_engine_not_a_fallback():
    """Without a device a rendering handle cannot be created (the driver's CPU box has no GPU);
    a dry handle refuses to render."""
    import torch
    if not torch.cuda.is_available():
        with pytest.raises(Exception):
            Runtime(44100.0, 512, device=0)
    rt = Runtime(44100.0, 512, device=-1)
    assert rt.render(el.cycle(440))["result"] == 0
    with pytest.raises(RuntimeError):
        rt.process(None, 1, 512)


def dry(sr=44100.0, bs=512):
    return Runtime(sr, bs, device=-1)


def test_return_codes_match_reference():
    """runtime/elem/Types.h:51-86 through Runtime.h:170-433 and the nodes' setProperty."""
    batches = [
        [[0, 1, "nope"]],                                   # 1 unknown node type
        [[0, 1, "const"], [0, 1, "const"]],                 # 3 node already exists
        [[3, 99, "value", 1]],                              # 2 node not found
        [[0, 1, "const"], [3, 1, "value", "hi"]],           # 5 invalid property type
        [[0, 1, "seq"], [3, 1, "offset", -1]],              # 6 invalid property value
        [[0, 1, "metro"], [3, 1, "interval", 0]],           # 6
        [[0, 1, "svf"], [3, 1, "mode", 3]],                 # 5
        [[0, 1, "sampleseq"], [3, 1, "duration", 0]],       # 6 (SampleSeq.h:183-190)
        [[0, 1, "sampleseq"], [3, 1, "duration", "x"]],     # 5
        [[0, 1, "sampleseq"], [3, 1, "path", "/missing"]],  # 6 (:192-203)
        [[0, 1, "sampleseq"], [3, 1, "seq", 4]],            # 5 (:205-209)
        [[0, 1, "table"], [3, 1, "path", "/missing"]],      # 6 (Table.h:22-27)
        [[0, 1, "seq2"], [3, 1, "offset", -1]],             # 6 (Seq2.h:54-61)
        [[0, 1, "sparseq2"], [3, 1, "interpolate", "x"]],   # 5 (SparSeq2.h:46-50)
        [[0, 1, "sample"], [3, 1, "startOffset", -3]],      # 6 (Sample.h:50-61)
        [[0, 1, "sample"], [3, 1, "mode", 2]],              # 5 (:37-47)
        [[0, 1, "scope"], [3, 1, "size", 100]],             # 6 (Analyzers.h:153-159)
        [[0, 1, "scope"], [3, 1, "name", 3]],               # 5 (:169-172)
        [[2, 1, 2, 0]],                                     # 2
        [[0, 1, "root"], [4, [1, 2]], [5]],                 # 2 (activateRoots on a missing node)
        ["x"],                                              # 8 invalid instruction format
        [[0, "a", "const"]],                                # 8
        [[7, 1, 2], [0, 5, "const"], [5]],                  # unknown opcodes are ignored -> 0
        [[0, 1, "root"], [0, 2, "const"], [2, 1, 2, 0], [3, 1, "channel", 0], [4, [1]], [5]],   # 0
    ]
    have_ref = oracle.have_ref()
    expect = [1, 3, 2, 5, 6, 6, 5, 6, 5, 6, 5, 6, 6, 5, 6, 5, 6, 5, 2, 2, 8, 8, 0, 0]
    for b, want in zip(batches, expect):
        got = dry().apply_instructions(b)
        assert got == want, (b, got)
        if oracle.have_port():
            assert oracle.PortRuntime(44100.0, 512).apply_instructions(b) == want, b
        if have_ref:
            assert oracle.RefRuntime(44100.0, 512).apply_instructions(b) == want, b


def test_plan_shape_for_the_benchmark_graph():
    """C2: one island per voice (all 13 ops fused behind LDS), the two 128-input mixers as islands on the second launch
    level, each cut into eight 64-frame runs over the waves of its two workgroups."""
    rt = dry(graphs.C2_SAMPLE_RATE)
    res = rt.render(*graphs.c2_graph())
    assert res["result"] == 0 and res["nodesAdded"] == 4107
    p = rt.describe_plan()
    assert p["num_nodes"] == 4107 and p["num_roots"] == 2 and p["num_levels"] == 2
    assert p["level_sizes"][0] == 256               # 256 voice workgroups
    assert p["level_sizes"][1] == 4                 # 2 mixers x 2 workgroups (4 active waves each)
    assert p["max_lds_bytes"] < 150 * 1024          # 6 pipelined buffer sets per voice island
    assert p["islands"][0]["copies"] == 6 and p["islands"][0]["stateless"] == 0
    assert p["num_hbm_buffers"] == 32 + 256 + 2     # host inputs + voice exports + roots


def test_lane_packing_plan_shapes():
    """Auto lane-packing: 256 voices = one island per CU, untouched; 512 voices = 2 per island (4 buffer sets: packed islands
    evaluate the svf coefficients inside the scan, no scratch); 1024 voices: still 2 per island (auto never goes beyond 2: deeper
    packs leave too few buffer sets and the block pipeline starves); 4 per island asked for explicitly: two buffer sets (without
    the fused coefficients only one would fit and the planner settles for 3: 342 islands); pack_islands = 1 switches it off.
    The two mixers stay as they are."""
    for voices, opts, want_k, want_islands in ((256, {}, 1, 256), (512, {}, 2, 256), (1024, {}, 2, 512), (1024, {"pack_islands": 4}, 4, 256),
                                               (1024, {"pack_islands": 4, "fuse_svf_coef": 0}, 3, 342), (512, {"pack_islands": 1}, 1, 512)):
        rt = dry(graphs.C2_SAMPLE_RATE)
        for k, v in opts.items():
            rt.set_option(k, v)
        assert rt.render(*graphs.c2_graph(voices=voices))["result"] == 0
        p = rt.describe_plan()
        assert p["pack_k"] == want_k and p["level_sizes"] == [want_islands, 4], (voices, p["pack_k"], p["level_sizes"])
        assert p["islands"][0]["copies"] >= 2 and p["max_lds_bytes"] <= 159 * 1024
        assert p["num_nodes"] == voices * 16 + 11


def _op_names():
    import os, re
    src = open(os.path.join(os.path.dirname(__file__), "..", "elementary_amd", "csrc", "device.h")).read()
    body = src[src.index("enum Op : uint16_t {"):src.index("OP_COUNT_")]
    return {i: t[3:].lower() for i, t in enumerate(re.findall(r"OP_[A-Z0-9_]+", body))}


def test_voice_island_schedule():
    """Planner rules visible in the per-wave task lists of a C2 voice island: the oscillator is two tasks (phase
    recurrence — merged with the gate phasor's — then the waveform with its consumers in the next stage), the filter-coefficient pre-pass shares
    a stage with its producers and is split over two waves, every recurrence has a wave to itself, and no stage
    order is violated inside a wave."""
    names = _op_names()
    rt = dry(graphs.C2_SAMPLE_RATE)
    assert rt.render(*graphs.c2_graph())["result"] == 0
    isl = rt.describe_plan()["islands"][0]
    waves = [[(names[o], st) for o, st in w] for w in isl["waves"]]
    flat = [t for w in waves for t in w]
    assert isl["stages"] == 6 and len(flat) == isl["tasks"]
    stage_of = lambda name: sorted({st for n, st in flat if n == name})
    # the gate phasor and both oscillator phases (constant frequencies) share ONE recurrence task (device.h OP_PHASE)
    assert stage_of("phase") == [0] and stage_of("saw_shape") == [1] and stage_of("blepsaw") == [] and stage_of("phasor") == []
    (coef_stage,) = stage_of("svf_coef")
    assert stage_of("svf") == [coef_stage + 1]
    coef_waves = [w for w in waves if ("svf_coef", coef_stage) in w]
    assert len(coef_waves) == 2 and all(w[-1][0] == "svf_coef" and len(w) == 3 for w in coef_waves)   # mul, add, coef
    for rec in ("phase", "pole", "svf"):
        assert [w for w in waves if any(n == rec for n, _ in w)] == [[(rec, stage_of(rec)[0])]], rec
    for w in waves:
        assert [st for _, st in w] == sorted(st for _, st in w)


def test_small_island_uses_spare_waves():
    """C1 (one 18-node island): 4 of its 8 waves would idle, so the heavy sample-parallel stage is cut four ways."""
    names = _op_names()
    rt = dry(graphs.C1_SAMPLE_RATE)
    assert rt.render(*graphs.c1_graph())["result"] == 0
    isl = rt.describe_plan()["islands"][0]
    coef = [w for w in isl["waves"] if any(names[o] == "svf_coef" for o, _ in w)]
    assert len(coef) == 4


def test_plan_handles_deep_and_wide_graphs():
    x = el.in_({"channel": 0})
    for k in range(300):                             # 300-deep chain -> several islands in sequence
        x = el.pole(0.5, el.mul(0.5, x))
    rt = dry()
    assert rt.render(x)["result"] == 0
    p = rt.describe_plan()
    assert p["num_levels"] >= 8 and p["max_lds_bytes"] <= 160 * 1024
    wide = el.add(*[el.cycle(100.0 + k) for k in range(500)])
    rt = dry()
    assert rt.render(wide)["result"] == 0
    assert rt.describe_plan()["num_nodes"] > 1500


def test_gc_follows_render_sequence_membership():
    """Runtime.h:220-272 / gc.test.js: nodes die only once no current or pending sequence holds them."""
    rts = [dry()]
    if oracle.have_port():
        rts.append(oracle.PortRuntime(44100.0, 512))
    results = []
    for rt in rts:
        rt.render(el.mul(2, 3))
        a = rt.gc()
        rt.render(el.mul(3, 4))       # old root keeps fading => still a current root => still held
        b = rt.gc()
        results.append((a, len(b)))
    assert all(r == results[0] for r in results)
    assert results[0][0] == []


def test_shared_resources_are_insert_only():
    rt = dry()
    assert rt.add_shared_resource("ir", np.ones(16, dtype=np.float32)) is True
    assert rt.add_shared_resource("ir", np.zeros(16, dtype=np.float32)) is False
    rt.prune_shared_resources()
    assert rt.add_shared_resource("ir", np.zeros(16, dtype=np.float32)) is True   # pruned: nobody held it


def test_multi_output_node_gets_one_plan_entry_per_channel():
    """GraphRenderSequence.h:15-24: a node's output buffers = highest outlet channel + 1. mc.table unpacked to 3 channels
    becomes three table lookups (one record each) feeding the add."""
    import numpy as np
    from elementary_amd import el
    from elementary_amd.runtime import Runtime
    rt = Runtime(44100.0, 512, device=-1)
    assert rt.add_shared_resource("/v/stereo", np.asarray([[27, 27, 27], [15, 15, 15]], np.float32))
    res = rt.render(el.add(*el.mc.table({"path": "/v/stereo", "channels": 3}, 0)))
    assert res["result"] == 0
    appends = [i for i in res["batch"] if i[0] == 2]
    table_id = [i[1] for i in res["batch"] if i[0] == 0 and i[2] == "mc.table"][0]
    assert sorted(i[3] for i in appends if i[2] == table_id) == [0, 1, 2]     # APPEND_CHILD carries the child's output channel
    plan = rt.describe_plan()
    assert plan["num_nodes"] == 4                                              # const, mc.table, add, root
    assert plan["num_tasks"] >= 5                                              # 3 table lookups + add + root


def test_process_does_not_wait_for_a_plan_build():
    """Runtime.h:207-216, 277-285: the reference hands a finished render sequence to the audio thread through an SPSC
    queue, so `process` never waits for `buildRenderSequence`. Here a commit drops the render lock while it plans: with
    the unlocked part of the build stretched to 0.5 s, `process` calls from another thread return at once (a dry handle
    answers 101 after taking the render lock), and the commit still lands."""
    import threading, time
    rt = dry(48000.0)
    assert rt.render(el.cycle(440.0))["result"] == 0
    rt.set_option("debug_build_delay_ms", 500)
    done = {}

    def commit():
        t0 = time.perf_counter()
        done["rc"] = rt.render(el.mul(0.5, el.cycle(220.0)), el.cycle(330.0))["result"]
        done["s"] = time.perf_counter() - t0

    th = threading.Thread(target=commit)
    th.start()
    time.sleep(0.1)     # the commit is inside its build now
    import gc
    gc.collect(); gc.disable()      # (a generation-2 collection of the test session's heap takes longer than the bound below)
    worst, calls = 0.0, 0
    while th.is_alive() and calls < 200:
        t0 = time.perf_counter()
        try:
            rt.process(None, 1, 512)
            code = None
        except RuntimeError as e:      # (not pytest.raises: its traceback capture costs more than the call being timed)
            code = str(e)
        worst = max(worst, time.perf_counter() - t0)
        assert code is not None and "101" in code
        calls += 1
        time.sleep(0.002)
    th.join()
    gc.enable()
    assert done["rc"] == 0 and done["s"] >= 0.5
    assert calls >= 20 and worst < 0.1, (calls, worst)
    assert rt.describe_plan()["num_roots"] == 2


def test_island_program_cache_reuses_unchanged_islands():
    """plan.cpp "island program cache": on the C5 mutation stream (one voice of 128 replaced per batch) a re-plan takes the
    unchanged voices' Island headers, program blobs and kernel texts from the previous build. `plan_cache` = 2 schedules every
    island anyway and compares it with the cached program byte for byte; = 1 must end with the same plan as = 0."""
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "benchmarks"))
    import bench_configs as B
    texts, _, _ = B._c5_batches(128, 24)
    plans = {}
    for mode in (0, 1, 2):
        rt = dry(graphs.C2_SAMPLE_RATE)
        rt.set_option("plan_cache", mode)
        rt.set_option("specialize", 2)
        for i, t in enumerate(texts):
            assert rt.apply_instructions_json(t) == 0
            if i % 16 == 15:
                rt.gc()
        plans[mode] = rt.describe_plan()
    assert plans[2]["plan_cache_mismatches"] == 0 and plans[2]["plan_islands_scheduled"] > 24 * 128
    # an island with the STRUCTURE of one scheduled before (another voice of the patch, the voice that replaces one) takes that
    # program with its records / arena buffers renamed; verify mode schedules it anyway and compares the two, bit for bit
    assert plans[2]["plan_relocation_mismatches"] == 0 and plans[2]["plan_islands_relocated"] >= 127 + 24
    assert plans[1]["plan_islands_relocated"] >= 127 + 24 and plans[1]["plan_islands_scheduled"] <= 8
    assert plans[1]["plan_islands_reused"] > 20 * 120                 # ~126 of 130 islands per re-plan
    strip = lambda p: {k: v for k, v in p.items() if not k.startswith("plan_") and k != "build_us"}      # noqa: E731
    assert strip(plans[1]) == strip(plans[0]) == strip(plans[2])
    assert plans[1]["plan_prog_heaps"] == 1                          # 24 re-plans of two islands each: the first heap still serves
    # island programs live in a device heap that is only ever appended to; one that keeps running out is replaced (and the
    # cache forgotten) without the plan changing
    rt = dry(graphs.C2_SAMPLE_RATE)
    rt.set_option("specialize", 2)
    rt.set_option("prog_heap_dwords", 290_000)                       # the 130 programs take ~270k dwords, a re-plan adds ~4k
    for i, t in enumerate(texts):
        assert rt.apply_instructions_json(t) == 0
        if i % 16 == 15:
            rt.gc()
    small = rt.describe_plan()
    assert strip(small) == strip(plans[1]) and small["plan_prog_heaps"] >= 4 and small["plan_prog_heap_used_dwords"] <= 290_000
    # other graph families through the comparing mode: every node case, re-rendered twice (second build: all hits)
    from cases import NODE_CASES, node_case_resources
    rt = dry(44100.0)
    rt.set_option("plan_cache", 2)
    for name, data in node_case_resources().items():
        assert rt.add_shared_resource(name, data)
    for name in sorted(NODE_CASES):
        roots = NODE_CASES[name][0]()
        assert rt.render(*roots)["result"] == 0
        assert rt.render(*roots[:1])["result"] == 0
        assert rt.render(*roots)["result"] == 0
    assert rt.describe_plan()["plan_cache_mismatches"] == 0


def test_block_sizes_above_one_lds_slot():
    """Runtime(sr, blockSize) (Runtime.h:44): up to 512 frames as given; above that (up to 32768), a size that splits into k equal slices
    of 64 .. 512 frames is rendered slice by slice (r04: multiples of 512 only), any other size — a prime — as slices of 512 frames and
    a shorter last one (refused until late r05); a graph with taps — whose loop delay is the host's block — commits as well (every slice
    works on its own stretch of host-block-sized tap buffers; r04 answered 104); above 32768 creation fails."""
    from elementary_amd.runtime import ElemHipError
    for bs in (1024, 2048, 32768, 700, 1000, 1023, 514, 521, 1031):
        rt = dry(48000.0, bs)
        assert rt.block_size == bs
        assert rt.render(el.mul(0.5, el.cycle(220.0)))["result"] == 0
        loop = el.tapOut({"name": "fb"}, el.add(el.in_({"channel": 0}), el.mul(0.5, el.tapIn({"name": "fb"}))))
        assert rt.render(loop)["result"] == 0
        assert rt.render(el.mul(0.25, el.cycle(330.0)))["result"] == 0
    for bs in (512 * 65, 0, -4):
        with pytest.raises(ElemHipError):
            dry(48000.0, bs)
    ok = dry(48000.0, 512)
    loop = el.tapOut({"name": "fb"}, el.add(el.in_({"channel": 0}), el.mul(0.5, el.tapIn({"name": "fb"}))))
    assert ok.render(loop)["result"] == 0


def _verify_mode(sr, renders, opts=()):
    rt = dry(sr)
    rt.set_option("specialize", 2)
    rt.set_option("plan_cache", 2)
    for k, v in opts:
        rt.set_option(k, v)
    from cases import node_case_resources
    for name, data in node_case_resources().items():
        rt.add_shared_resource(name, data)
    for roots in renders:
        assert rt.render(*roots)["result"] == 0
    d = rt.describe_plan()
    return d["plan_islands_relocated"], d["plan_relocation_mismatches"] + d["plan_cache_mismatches"]


def test_relocated_programs_in_verify_mode():
    """plan_cache = 2 over graph families with structural twins: an island that takes the renamed program of a twin (plan.cpp,
    "same structure, other nodes") is scheduled anyway and the two blobs compared. Every node case in ONE engine (twins across
    cases), 16 / 48 / 300 voices (unpacked and lane-packed), render jobs packed across roots, tap loops: no mismatch, and the path
    is really taken."""
    from cases import NODE_CASES
    total = 0
    for sr, renders, opts in ((44100.0, [NODE_CASES[n][0]() for n in sorted(NODE_CASES)], ()),
                              (graphs.C2_SAMPLE_RATE, [graphs.c2_graph(voices=v) for v in (16, 48)], ()),
                              (graphs.C2_SAMPLE_RATE, [graphs.c2_graph(voices=300)], (("cu_count", 150),)),
                              (graphs.C4_SAMPLE_RATE, [[graphs.c4_instance(k) for k in range(96)]], (("pack_roots", 1), ("cu_count", 32)))):
        relocated, bad = _verify_mode(sr, renders, opts)
        assert bad == 0, (sr, relocated, bad)
        total += relocated
    assert total >= 16 + 48 + 100


def test_release_library_carries_no_measurement_hooks():
    """the wrong-sample hooks (ELEMHIP_EXP_*, the out-of-range chain stores, the persistent chains) and the
    ELEMHIP_JIT_DEFINES back door exist only in a `make EXPERIMENTAL=1` build: a release library's embedded node-library text
    (what its run-time compiler sees) has the blocks removed (tools/strip_experimental.py), jit.cpp does not read the variable,
    `conv_mfma` takes 0 / 1, and the only tunings the compiler accepts are the whitelisted bit-identical ones."""
    from elementary_amd.runtime import LIB_PATH, ElemHipError
    data = open(LIB_PATH, "rb").read()
    for needle in (b"ELEMHIP_EXP_", b"ELEMHIP_JIT_DEFINES", b"ELEMHIP_CHAIN_OOB_STORES", b"ELEMHIP_PERSISTENT_CHAINS", b"ELEMHIP_STREAM_PER_BLOCK"):
        assert needle not in data, needle
    assert b"ELEMHIP_SPEC_BLOCK" in data            # (the embedded text is there)
    rt = Runtime(48000.0, 512, device=-1)
    rt.set_option("conv_mfma", 2)                    # clamped to 1: the operand-swapped bring-up probe is gone
    rt.set_option("biquad_form", 1)
    rt.set_option("biquad_form", 5)                  # (r06: forms 0-5, every one the reference's nine IEEE operations on the same operands)
    rt.set_option("biquad_form", 4)                  # ... back to the default
    with pytest.raises(ElemHipError):
        rt.set_option("biquad_form", 6)
    with pytest.raises(ElemHipError):
        rt.set_option("stream_ring", 0)
    # and the stripper itself: nested, #elif chains, #else branches
    import subprocess, sys, tempfile
    src = "a\n#if defined(ELEMHIP_EXPERIMENTAL) && defined(X)\nbad1\n#elif defined(ELEMHIP_EXPERIMENTAL) && defined(Y)\nbad2\n#else\nkeep1\n#endif\n" \
          "#ifdef OTHER\nkeep2\n#if defined(ELEMHIP_EXPERIMENTAL) && defined(Z)\nbad3\n#endif\n#else\nkeep3\n#endif\n" \
          "#if defined(ELEMHIP_EXPERIMENTAL) && defined(A)\nbad4\n#elif FOO\nkeep4\n#else\nkeep5\n#endif\nz\n"
    with tempfile.NamedTemporaryFile("w", suffix=".inc", delete=False) as f:
        f.write(src)
    out = subprocess.run([sys.executable, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "strip_experimental.py"), f.name],
                         capture_output=True, text=True, check=True).stdout
    os.unlink(f.name)
    assert out == "a\nkeep1\n#ifdef OTHER\nkeep2\n#else\nkeep3\n#endif\n#if FOO\nkeep4\n#else\nkeep5\n#endif\nz\n", out


def test_kernel_cache_is_bounded_and_one_off_shapes_are_deferred():
    """jit.cpp (r05): the in-memory table is capped (entries no plan references are evicted, least recently used first), an entry
    keeps a few KB of generated text, not the 300 KB translation unit; describe_plan() reports the compiler's books. (In a process
    of its own: the table is process-wide, and this suite's other engines leave entries behind.)"""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r"""
import json, sys
sys.path.insert(0, %r)
import torch
from elementary_amd import el
from elementary_amd.runtime import Runtime
rt = Runtime(44100.0, 512, device=-1)
rt.set_option("specialize", 2)
rt.set_option("jit_cache_entries", 4)
x = el.in_({"channel": 0})
ops = [el.tanh, el.sin, lambda s: el.mul(0.5, s), lambda s: el.add(0.1, s), el.abs, lambda s: el.pole(0.5, s), el.cos]
for k, op in enumerate(ops):        # seven structurally different one-island graphs, one after the other
    assert rt.render(op(el.lowpass(500.0 + k, 0.7, x)))["result"] == 0
print(json.dumps(rt.describe_plan()))
""" % root
    res = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-2000:]
    p = json.loads(res.stdout.strip().splitlines()[-1])
    jit = p["jit"]
    assert jit["entries"] <= 4 + 2 and jit["evictions"] >= 1, jit
    assert jit["compiles"] + jit["disk_hits"] >= 7, jit
    assert jit["text_bytes_held"] <= jit["entries"] * 200_000, jit          # generated text only
    assert p["shapes"]["total"] >= 1 and p["shapes"]["ready"] == p["shapes"]["total"]
    assert 0.0 <= p["interp_block_fraction"] <= 1.0


def test_kernel_cache_trusts_only_a_directory_of_its_own(tmp_path):
    """the on-disk kernel cache holds code this process will RUN. (a) A cache directory other users may write to
    is not used at all: the shape still compiles (the helper works in a private scratch directory) but nothing is read from or
    published to the directory. (b) In a trusted directory a file that is not a gfx code object, and a symlink of the right name,
    are ignored and replaced by a real compile; compile scratch directories do not stay behind; published files are ELF objects
    for EM_AMDGPU that nobody else may write."""
    import json
    import stat
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r"""
import json, sys
sys.path.insert(0, %r)
import torch
from elementary_amd import el
from elementary_amd.runtime import Runtime
rt = Runtime(44100.0, 512, device=-1)
rt.set_option("specialize", 2)
assert rt.render(el.tanh(el.lowpass(432.0, 0.7, el.in_({"channel": 0}))))["result"] == 0
print(json.dumps(rt.describe_plan()))
""" % root

    def run(kc):
        res = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=dict(os.environ, ELEMHIP_KCACHE=str(kc)))
        assert res.returncode == 0, res.stderr[-2000:]
        return json.loads(res.stdout.strip().splitlines()[-1]), res.stderr

    # (a) a world-writable directory
    open_dir = tmp_path / "open"
    open_dir.mkdir()
    os.chmod(open_dir, 0o777)
    p, err = run(open_dir)
    assert p["shapes"]["ready"] == p["shapes"]["total"] >= 1 and p["jit"]["compiles"] >= 1 and p["jit"]["disk_hits"] == 0
    assert "on-disk kernel cache is off" in err
    assert [f for f in os.listdir(open_dir)] == []
    # (b) a directory of our own: first run compiles and publishes
    own = tmp_path / "own"
    p, _ = run(own)
    assert stat.S_IMODE(os.stat(own).st_mode) == 0o700
    files = sorted(os.listdir(own))
    assert files and all(f.endswith(".hsaco") for f in files), files          # no scratch directory, no temp file left behind
    for f in files:
        raw = open(own / f, "rb").read()
        assert raw[:4] == b"\x7fELF" and raw[4] == 2 and int.from_bytes(raw[18:20], "little") == 224
        assert stat.S_IMODE(os.stat(own / f).st_mode) & 0o022 == 0
    # second run: disk hits
    p2, _ = run(own)
    assert p2["jit"]["disk_hits"] >= 1 and p2["jit"]["compiles"] == 0
    # a planted file of the right name that is not a code object, then a symlink of the right name: both ignored, compiled afresh
    victim = own / files[0]
    good = open(victim, "rb").read()
    open(victim, "wb").write(b"#!/bin/sh\necho not a code object\n" * 40)
    p3, _ = run(own)
    assert p3["jit"]["compiles"] >= 1 and p3["shapes"]["ready"] == p3["shapes"]["total"]
    assert open(victim, "rb").read()[:4] == b"\x7fELF"
    os.remove(victim)
    elsewhere = tmp_path / "elsewhere.hsaco"
    open(elsewhere, "wb").write(good)
    os.symlink(elsewhere, victim)
    p4, _ = run(own)
    assert p4["jit"]["compiles"] >= 1                                       # the link was not followed as a cache hit
    assert not os.path.islink(victim)                                       # ... and the publish replaced it with a file of our own


def test_fft4096_core_on_the_host():
    """elementary_amd/csrc/fft4096.h (the transform core of the long-partition convolver, conv_long.inc) compiled for the HOST:
    256 emulated threads, forward / inverse real transforms of 8192 samples against a double-precision DFT and one overlap-save
    convolution step against the direct sum."""
    import json
    import shutil
    import subprocess
    import tempfile
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cxx = "/opt/rocm/lib/llvm/bin/clang++" if os.path.exists("/opt/rocm/lib/llvm/bin/clang++") else shutil.which("clang++")
    if not cxx:
        pytest.skip("needs clang++ (ext_vector_type)")
    with tempfile.TemporaryDirectory() as d:
        exe = os.path.join(d, "fft4096_host")
        subprocess.run([cxx, "-std=c++17", "-O2", "-ffp-contract=off", "-I", os.path.join(root, "elementary_amd", "csrc"),
                        os.path.join(root, "tests", "native", "fft4096_host.cpp"), "-o", exe], check=True)
        res = subprocess.run([exe], capture_output=True, text=True)
    out = json.loads(res.stdout.strip().splitlines()[-1])
    assert res.returncode == 0 and out["ok"], out
    assert out["overlap_save_max_err"] <= 5e-7 and out["roundtrip_max_err"] <= 2e-6, out


def test_rccl_exchange_snippet_compiles_and_links():
    """INTEGRATION.md §6 / north_star "RCCL over xGMI only for the final output-bus reduce": the documented exchange step (RCCL
    send / recv to rank 0 + elemhip_sum_buses in rank order, and the ncclReduce alternative) as a translation unit compiled and
    linked against librccl and libelemhip. Not run: no multi-GPU node is available to the builder."""
    import shutil
    import subprocess
    import tempfile
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cxx = "/opt/rocm/lib/llvm/bin/clang++" if os.path.exists("/opt/rocm/lib/llvm/bin/clang++") else shutil.which("hipcc")
    if not cxx or not os.path.exists("/opt/rocm/include/rccl/rccl.h"):
        pytest.skip("needs the ROCm toolchain and the RCCL headers")
    with tempfile.TemporaryDirectory() as d:
        res = subprocess.run([cxx, "-std=c++17", "-O1", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-I" + os.path.join(root, "include"),
                              os.path.join(root, "tests", "native", "rccl_bus_sum.cpp"), "-L" + os.path.join(root, "elementary_amd"), "-lelemhip",
                              "-L/opt/rocm/lib", "-lrccl", "-lamdhip64", "-Wl,-rpath,/opt/rocm/lib", "-o", os.path.join(d, "rccl_bus_sum")],
                             capture_output=True, text=True)
        assert res.returncode == 0, res.stderr[-2000:]
        assert os.path.getsize(os.path.join(d, "rccl_bus_sum")) > 1000


def test_round5_call_path_options_on_a_dry_handle():
    """`sync_poll` (elemhip_process ends on the epilogue's word in mapped host memory), `resident` / `resident_idle_us` / `resident_after`
    (the opt-in resident kernel) and `spec_waves_per_eu` (register cap of the generated kernels) are options every handle takes, and
    `elemhip_describe_plan` says which way a handle waits; unknown keys are still refused; the stats struct carries the resident counters."""
    from elementary_amd.runtime import ElemHipError
    rt = dry(48000.0)
    assert rt.render(el.mul(0.5, el.cycle(220.0)))["result"] == 0
    d = rt.describe_plan()
    assert d["sync_poll"] == 1 and d["resident"] == 0 and d["sync_polls"] == 0 and d["resident_blocks"] == 0
    for key, val in (("sync_poll", 0), ("resident", 1), ("resident_idle_us", 500), ("resident_after", 5), ("spec_waves_per_eu", 4)):
        rt.set_option(key, val)
    d = rt.describe_plan()
    assert d["sync_poll"] == 0 and d["resident"] == 1
    st = rt.stats()
    assert st["resident_launches"] == 0 and st["resident_blocks"] == 0
    with pytest.raises(ElemHipError):
        rt.set_option("resident_kernel", 1)
    # the register cap is part of a generated kernel's text (and so of its cache key), nothing else changes
    rt2 = dry(48000.0)
    rt2.set_option("specialize", 2)
    assert rt2.render(el.mul(0.5, el.cycle(220.0)))["result"] == 0
    plain = rt2.spec_info(0)["source"]
    rt2.set_option("spec_waves_per_eu", 4)
    assert rt2.render(el.mul(0.25, el.cycle(221.0)))["result"] == 0
    capped = rt2.spec_info(0)["source"]
    assert "amdgpu_waves_per_eu(4, 4)" in capped and "amdgpu_waves_per_eu" not in plain
