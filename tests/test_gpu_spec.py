"""Run-time specialised island kernels (elementary_amd/csrc/jit.cpp, codegen.cpp, island_spec.inc) vs the reference
engine and vs the interpreter kernels. Every test forces `specialize = 2` (commit waits for the compiler), renders
through elemhip_process_blocks (the only path that uses the specialised kernels) and checks that they actually ran.
Tolerance 1e-6 absolute (x max|ref| when > 1); the interpreter and the specialised kernels share their op bodies, so
everything that is not a libm transcendental is expected to be bit-identical between the two."""
import numpy as np
import pytest

from elementary_amd import el, graphs
from helpers import lcg_noise
from cases import NODE_CASES, REF_ONLY, node_case_resources

pytestmark = pytest.mark.gpu
TOL = 1e-6


def _checker(sr, bs):
    import oracle
    return oracle.RefRuntime(sr, bs) if oracle.have_ref() else oracle.PortRuntime(sr, bs)


def _spec_runtime(sr, bs, batch=None, copies=None):
    from elementary_amd.runtime import Runtime
    rt = Runtime(sr, bs, device=0)
    rt.set_option("specialize", 2)
    if batch is not None:
        rt.set_option("batch_blocks", batch)
    if copies is not None:
        rt.set_option("pipeline_copies", copies)
    return rt


def _render_blocks(rt, nb, n_out, x=None, block=512):
    import torch
    out = torch.zeros((nb, n_out, block), dtype=torch.float32, device="cuda")
    if x is not None:
        xin = torch.from_numpy(x).cuda()
        torch.cuda.synchronize()
        rt.process_blocks(nb, n_out, out_ptr=out.data_ptr(), in_ptr=xin.data_ptr(), num_inputs=x.shape[1])
    else:
        torch.cuda.synchronize()
        rt.process_blocks(nb, n_out, out_ptr=out.data_ptr())
    return out.cpu().numpy()


def _assert_ran_specialised(rt):
    """Unconditional wherever the plan has an island program (the planner writes a specialised kernel for every island when
    the block is a multiple of 64 frames — stateful ones, tap islands, and since r04 the stateless ones too: mixers, root
    gains, pure math): the shape must exist, be compiled and have been launched."""
    st = rt.stats()
    plan = rt.describe_plan()
    expect = (rt.block_size % 64 == 0 and any(i["tasks"] > 0 for i in plan["islands"]))
    if not expect:
        assert st["spec_shapes"] == 0, st
        return
    assert st["spec_shapes"] > 0 and st["batch_launches"] > 0, st
    for k in range(st["spec_shapes"]):
        info = rt.spec_info(k)
        assert info["state"] == 1, info["log"][:2000]
    assert st["spec_launches"] > 0, st


@pytest.mark.parametrize("name", sorted(NODE_CASES))
def test_spec_node_case(gpu_required, name):
    """Every node case: specialised multi-block launches (5 blocks per launch, 17 blocks) vs the reference engine block by
    block, and vs the interpreter kernels of a second HIP engine."""
    from elementary_amd.runtime import Runtime
    roots_fn, n_in = NODE_CASES[name]
    if name in REF_ONLY:
        import oracle
        if not oracle.have_ref():
            pytest.skip("needs oracle/_ref")
    nb = 17
    a, b, c = _spec_runtime(44100.0, 512, batch=5), Runtime(44100.0, 512), _checker(44100.0, 512)
    b.set_option("specialize", 0)
    for rt in (a, b, c):
        for rname, data in node_case_resources().items():
            assert rt.add_shared_resource(rname, data)
    roots = roots_fn()
    n_out = len(roots)
    for rt in (a, b, c):
        assert rt.render(*roots)["result"] == 0
    x = np.stack([np.stack([lcg_noise(512, 1 + ch + 97 * k, 0.5) for ch in range(max(n_in, 1))]) for k in range(nb)])
    got = _render_blocks(a, nb, n_out, x)
    interp = np.stack([b.process(x[k], n_out, 512) for k in range(nb)])
    ref = np.stack([c.process(x[k] if n_in else None, n_out, 512) for k in range(nb)])
    _assert_ran_specialised(a)
    scale = max(1.0, float(np.abs(ref).max()))
    assert np.isfinite(got).all()
    assert float(np.abs(got - ref).max()) <= TOL * scale, f"{name}: specialised vs reference {np.abs(got - ref).max():.3e}"
    assert float(np.abs(got - interp).max()) <= TOL * scale, f"{name}: specialised vs interpreter {np.abs(got - interp).max():.3e}"


def test_spec_c2_full_graph(gpu_required):
    """BASELINE configs[1] at full size through the specialised kernels: 4107 nodes, 256 voices = one island shape, 200 blocks."""
    a, c = _spec_runtime(graphs.C2_SAMPLE_RATE, 512), _checker(graphs.C2_SAMPLE_RATE, 512)
    roots = graphs.c2_graph()
    assert a.render(*roots)["result"] == 0 and c.render(*roots)["result"] == 0
    st = a.stats()
    assert st["spec_shapes"] == 2 and st["spec_islands"] >= 256 + 2, st      # the voice shape + the shape of the mixers / root gains
    got = _render_blocks(a, 200, 2)
    ref = np.stack([c.process(None, 2, 512) for _ in range(200)])
    _assert_ran_specialised(a)
    err = float(np.abs(got - ref).max())
    assert err <= TOL, f"C2 specialised: max abs err {err:.3e}, max|ref| {np.abs(ref).max():.3f}"


def test_spec_c4_instances(gpu_required):
    a, c = _spec_runtime(graphs.C4_SAMPLE_RATE, 512), _checker(graphs.C4_SAMPLE_RATE, 512)
    roots = [graphs.c4_instance(k) for k in range(8)]
    assert a.render(*roots)["result"] == 0 and c.render(*roots)["result"] == 0
    got = _render_blocks(a, 60, 8)
    ref = np.stack([c.process(None, 8, 512) for _ in range(60)])
    _assert_ran_specialised(a)
    assert float(np.abs(got - ref).max()) <= TOL


@pytest.mark.parametrize("copies", [1, 3, 6])
def test_spec_every_stateful_node_pipelined(gpu_required, copies):
    """One graph with every stateful node type, host inputs and time-dependent nodes, through specialised launches with
    1, 3 and 6 blocks in flight: equal to the interpreter's block-by-block rendering (same op bodies)."""
    from elementary_amd.runtime import Runtime
    from cases import every_stateful_roots as roots
    nb = 53
    x = np.stack([np.stack([lcg_noise(512, 7 + k, 0.5)]) for k in range(nb)])
    a, b = _spec_runtime(48000.0, 512, batch=12, copies=copies), Runtime(48000.0, 512)
    b.set_option("specialize", 0)
    assert a.render(*roots())["result"] == 0 and b.render(*roots())["result"] == 0
    got = _render_blocks(a, nb, 3, x)
    ref = np.stack([b.process(x[k], 3, 512) for k in range(nb)])
    _assert_ran_specialised(a)
    scale = max(1.0, float(np.abs(ref).max()))
    assert float(np.abs(got - ref).max()) <= TOL * scale, float(np.abs(got - ref).max())


@pytest.mark.parametrize("seed", range(0, 48, 3))
def test_spec_random_graph(gpu_required, seed):
    from test_gpu_fuzz import random_graph
    nb, n_out = 22, min(3, 1 + seed % 5)
    x = np.stack([np.stack([lcg_noise(512, 11 + 7 * k + ch, 0.5) for ch in range(2)]) for k in range(nb)])
    a, c = _spec_runtime(48000.0, 512, batch=5 + seed % 7), _checker(48000.0, 512)
    for rt in (a, c):
        assert rt.render(*random_graph(seed, n_nodes=24 + 22 * (seed % 4), n_roots=1 + seed % 5)[:n_out])["result"] == 0
    got = _render_blocks(a, nb, n_out, x)
    ref = np.stack([c.process(x[k], n_out, 512) for k in range(nb)])
    _assert_ran_specialised(a)
    scale = max(1.0, float(np.abs(ref).max()))
    assert float(np.abs(got - ref).max()) <= TOL * scale, f"seed {seed}: {np.abs(got - ref).max():.3e}"


def test_spec_falls_back_while_compiling(gpu_required):
    """specialize = 1: the first launches may go through the interpreter kernels, later ones through the specialised kernel;
    the stream of samples is the same either way (state lives in the node records, not in the kernel)."""
    from elementary_amd.runtime import Runtime
    a, c = Runtime(48000.0, 512, device=0), _checker(48000.0, 512)
    a.set_option("specialize", 1)
    roots = graphs.c2_graph(voices=16)
    assert a.render(*roots)["result"] == 0 and c.render(*roots)["result"] == 0
    got = np.concatenate([_render_blocks(a, 24, 2) for _ in range(12)])
    ref = np.stack([c.process(None, 2, 512) for _ in range(24 * 12)])
    assert float(np.abs(got - ref).max()) <= TOL


@pytest.mark.parametrize("bs", [64, 192, 256])
def test_spec_other_block_sizes(gpu_required, bs):
    """Specialised kernels at block sizes other than 512 (multiples of 64: the unit of their vector accesses): the
    recurrence loops pick their prefetch depth from the block length, the stage ranges shrink with it. 16 C2 voices and the
    every-stateful-node graph, 40 blocks each, vs the reference engine."""
    from cases import every_stateful_roots
    for roots_fn, n_out, n_in in ((lambda: graphs.c2_graph(voices=16), 2, 0), (every_stateful_roots, 3, 1)):
        a, c = _spec_runtime(48000.0, bs, batch=16), _checker(48000.0, bs)
        assert a.render(*roots_fn())["result"] == 0 and c.render(*roots_fn())["result"] == 0
        nb = 40
        x = np.stack([np.stack([lcg_noise(bs, 7 + k, 0.5)]) for k in range(nb)]) if n_in else None
        got = _render_blocks(a, nb, n_out, x, block=bs)
        ref = np.stack([c.process(x[k] if n_in else None, n_out, bs) for k in range(nb)])
        _assert_ran_specialised(a)
        scale = max(1.0, float(np.abs(ref).max()))
        assert float(np.abs(got - ref).max()) <= TOL * scale, (bs, float(np.abs(got - ref).max()))


@pytest.mark.parametrize("voices", [8, 48])
def test_process_uses_the_specialised_kernels_block_by_block(gpu_required, voices):
    """elemhip_process (Runtime::process, Runtime.h:274-291): once the root fades have settled and every island shape is
    compiled, a whole block is a launch set of one through the specialised kernels; partial blocks, and everything with
    `spec_blocks = 0`, keep to the interpreter kernels. All three agree with the reference engine and the two engines with
    each other bit for bit (shared op bodies)."""
    a, b, c = _spec_runtime(graphs.C2_SAMPLE_RATE, 512), _spec_runtime(graphs.C2_SAMPLE_RATE, 512), _checker(graphs.C2_SAMPLE_RATE, 512)
    b.set_option("spec_blocks", 0)
    roots = graphs.c2_graph(voices=voices)
    for rt in (a, b, c):
        assert rt.render(*roots)["result"] == 0
    sizes = [512] * 40 + [200, 312] + [512] * 10
    before = None
    for k, n in enumerate(sizes):
        ya, yb, yc = a.process(None, 2, n), b.process(None, 2, n), c.process(None, 2, n)
        assert float(np.abs(ya - yc).max()) <= TOL * max(1.0, float(np.abs(yc).max())), k
        assert np.array_equal(ya, yb), k
        if k == 30:
            before = a.stats()["spec_launches"]
    assert before is not None and a.stats()["spec_launches"] >= before + 19        # blocks 31..39 and 42..51
    assert b.stats()["spec_launches"] == 0


def _fuse_graphs():
    x = el.in_({"channel": 0})
    six_roots = [el.mul(0.1 * (k + 1), el.cycle(110.0 * (k + 1))) for k in range(6)]     # (render() roots: channel = position)
    return {
        "c1": (graphs.C1_SAMPLE_RATE, graphs.c1_graph, 0),
        "c2x16": (graphs.C2_SAMPLE_RATE, lambda: graphs.c2_graph(voices=16), 0),
        "six_channels": (44100.0, lambda: six_roots, 0),
        "filters_on_input": (44100.0, lambda: [el.lowpass(800.0, 1.2, x), el.add(el.pole(0.99, x), el.mul(0.3, el.cycle(330.0)))], 1),
    }


@pytest.mark.parametrize("name", ["c1", "c2x16", "six_channels", "filters_on_input"])
def test_fused_epilogue_of_a_single_block_call(gpu_required, name):
    """elemhip_process on a settled, fully compiled sequence: the last level's kernel ends with the epilogue (its last workgroup
    sums the output bus and advances the clock: island_spec.inc spec_epilogue_tail) instead of a launch of its own. Same adds in
    the same order: bit-identical to the separate epilogue launch, <= 1e-6 from the reference engine; a re-render in the middle
    (root fades: the engine falls back to block-at-a-time until they settle) and a different output count are part of the run."""
    from elementary_amd.runtime import Runtime
    sr, mk, n_in = _fuse_graphs()[name]
    roots = mk()
    n_out = len(roots)
    outs = {}
    for fuse in (1, 0):
        rt = Runtime(sr, 512, device=0)
        rt.set_option("specialize", 2); rt.set_option("fuse_epilogue", fuse)
        assert rt.render(*roots)["result"] == 0
        ys = []
        for k in range(40):
            xin = np.stack([lcg_noise(512, 11 + k, 0.5)]) if n_in else None
            if k == 22:
                assert rt.render(*roots[::-1])["result"] == 0          # roots swap channels: old ones fade out, new ones fade in
            ys.append(rt.process(xin, n_out + (1 if 30 <= k < 34 else 0), 512)[:n_out])
        outs[fuse] = np.stack(ys)
        plan = rt.describe_plan()
        fused = plan["plan_fused_epilogues"]
        assert (fused >= 20) if fuse else (fused == 0), fused
        # r06: the fused tail publishes the call's completion word itself — a fused call ends on the polled word like any other
        assert plan["sync_polls"] >= 38 and plan["sync_poll_fallbacks"] == 0, (plan["sync_polls"], plan["sync_poll_fallbacks"])
    assert np.array_equal(outs[1], outs[0])
    c = _checker(sr, 512)
    assert c.render(*roots)["result"] == 0
    ref = []
    for k in range(40):
        xin = np.stack([lcg_noise(512, 11 + k, 0.5)]) if n_in else None
        if k == 22:
            assert c.render(*roots[::-1])["result"] == 0
        ref.append(c.process(xin, n_out + (1 if 30 <= k < 34 else 0), 512)[:n_out])
    ref = np.stack(ref)
    assert float(np.abs(outs[1] - ref).max()) <= TOL * max(1.0, float(np.abs(ref).max()))


def _phasors_only_roots():
    ph = [el.phasor(f) for f in (1000.0, 700.0, 333.3)]
    return [ph[0], el.mul(2.0, ph[1]), el.le(ph[2], 0.5), el.sample({"path": "/t/ramp", "mode": "trigger", "startOffset": 10}, el.le(ph[1], 0.5), 2.0)]


@pytest.mark.parametrize("path", ["process", "sets_of_3"])
def test_merged_phase_task_of_phasors_only(gpu_required, path):
    """Constant-frequency phasors share one lane-per-node task (OP_PHASE) whose spare lanes repeat the last member. With no
    oscillator behind the phasors those lanes once ran the oscillator's arithmetic (inc = f / sr, not f * (1 / sr)) and raced
    their final phase into the phasor's record: every launch started 512 x ulp(inc) off (r04; 1000 Hz drifted 4e-4 in five
    blocks through elemhip_process, and a 700 Hz train — period exactly 63 frames — moved its edges). A phasor with a constant
    frequency is exact arithmetic: bit-identical to the reference engine once the root fades have settled, over many launches."""
    a, c = _spec_runtime(44100.0, 512, batch=3), _checker(44100.0, 512)
    roots = _phasors_only_roots()
    for rt in (a, c):
        for rname, data in node_case_resources().items():
            assert rt.add_shared_resource(rname, data)
        assert rt.render(*roots)["result"] == 0
    nb = 30
    ref = np.stack([c.process(None, 4, 512) for _ in range(nb)])
    got = np.stack([a.process(None, 4, 512) for _ in range(nb)]) if path == "process" else np.concatenate([_render_blocks(a, 3, 4) for _ in range(nb // 3)])
    assert a.stats()["spec_launches"] > 0
    assert np.array_equal(got[3:, :3], ref[3:, :3])                       # the phasors, their scaled copy and the train
    assert float(np.abs(got - ref).max()) <= TOL


@pytest.mark.parametrize("name", sorted(NODE_CASES))
def test_spec_node_case_call_by_call(gpu_required, name):
    """Every node case through elemhip_process, one synchronous call per block (the reference hosts' protocol): the two blocks
    of the root fade-in go through the specialised level launches + the per-block epilogue, the rest as launch sets of one — node
    state crosses a launch boundary after every block (records written back and staged again), which the multi-block test above
    does three times in 17 blocks. Against the reference engine and, bit for bit where no libm call is involved, the interpreter."""
    from elementary_amd.runtime import Runtime
    roots_fn, n_in = NODE_CASES[name]
    if name in REF_ONLY:
        import oracle
        if not oracle.have_ref():
            pytest.skip("needs oracle/_ref")
    nb = 12
    a, c = Runtime(44100.0, 512, device=0), _checker(44100.0, 512)
    a.set_option("specialize", 2)
    for rt in (a, c):
        for rname, data in node_case_resources().items():
            assert rt.add_shared_resource(rname, data)
    roots = roots_fn()
    n_out = len(roots)
    for rt in (a, c):
        assert rt.render(*roots)["result"] == 0
    for k in range(nb):
        x = np.stack([lcg_noise(512, 1 + ch + 97 * k, 0.5) for ch in range(max(n_in, 1))])
        got, ref = a.process(x if n_in else None, n_out, 512), c.process(x if n_in else None, n_out, 512)
        scale = max(1.0, float(np.abs(ref).max()))
        assert float(np.abs(got - ref).max()) <= TOL * scale, f"{name}: block {k}: {np.abs(got - ref).max():.3e}"
    info = a.describe_plan()
    if any(i["tasks"] > 0 for i in info["islands"]):
        assert a.stats()["spec_launches"] > 0 and info["plan_spec_fade_blocks"] >= 1, (a.stats(), info["plan_spec_fade_blocks"])


@pytest.mark.parametrize("seed", range(0, 48, 3))
def test_spec_random_graph_call_by_call(gpu_required, seed):
    from elementary_amd.runtime import Runtime
    from test_gpu_fuzz import random_graph
    nb, n_out = 14, min(3, 1 + seed % 5)
    a, c = Runtime(48000.0, 512, device=0), _checker(48000.0, 512)
    a.set_option("specialize", 2)
    for rt in (a, c):
        assert rt.render(*random_graph(seed, n_nodes=24 + 22 * (seed % 4), n_roots=1 + seed % 5)[:n_out])["result"] == 0
    for k in range(nb):
        x = np.stack([lcg_noise(512, 11 + 7 * k + ch, 0.5) for ch in range(2)])
        got, ref = a.process(x, n_out, 512), c.process(x, n_out, 512)
        scale = max(1.0, float(np.abs(ref).max()))
        assert float(np.abs(got - ref).max()) <= TOL * scale, f"seed {seed}: block {k}: {np.abs(got - ref).max():.3e}"
    assert a.stats()["spec_launches"] > 0


def test_register_capped_kernels_render_the_same_samples(gpu_required):
    """`spec_waves_per_eu` = 4 compiles the specialised kernels for 128 VGPRs (two eight-wave workgroups per CU, with
    `pipeline_copies` = 3 an island also fits half a CU's LDS): a measurement option (profiles/r05/occupancy_sweep_two_workgroups_per_cu.txt:
    slower than lane-packing at every size), and register allocation must not change a sample — the every-stateful-node graph and 24
    synth voices, bit for bit against the default kernels."""
    from cases import every_stateful_roots
    for roots_fn, n_out, n_in in ((every_stateful_roots, 3, 1), (lambda: graphs.c2_graph(voices=24), 2, 0)):
        outs = []
        for opts in ({}, {"spec_waves_per_eu": 4, "pipeline_copies": 3}):
            rt = _spec_runtime(48000.0, 512, batch=16)
            for k, v in opts.items():
                rt.set_option(k, v)
            assert rt.render(*roots_fn())["result"] == 0
            x = np.stack([np.stack([lcg_noise(512, 3 + b, 0.5) for _ in range(n_in)]) for b in range(48)]) if n_in else None
            outs.append(_render_blocks(rt, 48, n_out, x))
            assert rt.stats()["spec_launches"] > 0
        assert np.array_equal(outs[0], outs[1])
