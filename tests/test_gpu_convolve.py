"""`convolve` on the HIP engine (conv.hip) against (i) outputs recorded from the reference's wasm
engine, (ii) float64 linear convolution, (iii) the CPU restatement, at <= 1e-6 abs (|y| < 1)."""
import numpy as np
import pytest

import conv_cases as C
import oracle
from elementary_amd import graphs

TOL = 1e-6
pytestmark = pytest.mark.gpu


def hip(sr, bs):
    from elementary_amd.runtime import Runtime
    return Runtime(sr, bs, device=0)


@pytest.mark.parametrize("name", sorted(C.SCENARIOS))
def test_scenario_matches_reference_recording(gpu_required, name):
    y = C.run_scenario(hip, name).astype(np.float64)
    g = C.golden(name).astype(np.float64)
    assert y.shape == g.shape
    assert float(np.abs(y - g).max()) <= TOL, float(np.abs(y - g).max())
    assert float(np.abs(y - C.exact_model(name)).max()) <= TOL


def _c3(make, channels, blocks, block=512):
    rt = make(graphs.C3_SAMPLE_RATE, block)
    for ch in range(channels):
        assert rt.add_shared_resource(f"ir{ch}", graphs.c3_impulse_response(ch))
    assert rt.render(*graphs.c3_graph(channels))["result"] == 0
    x = graphs.c3_input(channels, blocks * block)
    return rt, x


def test_c3_eight_channels_vs_restatement(gpu_required):
    """BASELINE configs[2]: 8 channels x 96 000-tap IR, 64 blocks (8 full tail periods of the reference)."""
    outs = []
    for make in (hip, lambda sr, bs: oracle.PortRuntime(sr, bs)):
        rt, x = _c3(make, 8, 64)
        outs.append(np.stack([rt.process(x[:, k * 512:(k + 1) * 512], 8, 512) for k in range(64)]))
    assert float(np.abs(outs[1]).max()) > 0.2
    assert float(np.abs(outs[0].astype(np.float64) - outs[1]).max()) <= TOL


def test_c3_process_blocks_matches_process(gpu_required):
    """the multi-block path (conv.hip batch kernels: all partitions summed in one pass) renders the same samples as
    block-at-a-time process() (helpers' partial sums + main) up to the summation order."""
    import torch
    ch, blocks = 2, 24
    rt, x = _c3(hip, ch, blocks)
    ref = np.stack([rt.process(x[:, k * 512:(k + 1) * 512], ch, 512) for k in range(blocks)])
    rt2, _ = _c3(hip, ch, blocks)
    xin = torch.from_numpy(np.ascontiguousarray(x.reshape(ch, blocks, 512).transpose(1, 0, 2))).cuda()
    out = torch.empty((blocks, ch, 512), dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    rt2.process_blocks(blocks, ch, out_ptr=out.data_ptr(), in_ptr=xin.data_ptr(), num_inputs=ch)
    assert float(np.abs(out.cpu().numpy() - ref).max()) <= TOL


def test_convolve_graph_shapes(gpu_required):
    """The planner folds `in` leaves and roots into the convolve launch only when nothing else needs them; every
    shape must render like the CPU engine: folded both ways, `in` shared with another consumer, processed input,
    processed output, one convolver feeding two roots, a convolver behind a channel the host does not supply."""
    from elementary_amd import el
    ir = graphs.c3_impulse_response(0, 3000)
    x0, x1, x5 = el.in_({"channel": 0}), el.in_({"channel": 1}), el.in_({"channel": 5})
    shared = el.convolve({"path": "ir", "key": "shared"}, x1)
    roots = [el.convolve({"path": "ir", "key": "a"}, x0),                       # root(convolve(in)): both folded
             el.add(el.convolve({"path": "ir", "key": "b"}, x0), el.mul(0.25, x0)),   # `in` also feeds a mul: stays a task
             el.convolve({"path": "ir", "key": "c"}, el.mul(0.5, x1)),            # processed input
             el.mul(0.5, el.convolve({"path": "ir", "key": "d"}, x1)),            # processed output
             shared, el.tanh(shared),                                             # two consumers: root not folded
             el.convolve({"path": "ir", "key": "e"}, x5)]                        # channel 5 of 2: silence in
    outs = []
    for make in (hip, lambda sr, bs: oracle.PortRuntime(sr, bs)):
        rt = make(48000.0, 512)
        assert rt.add_shared_resource("ir", ir)
        assert rt.render(*roots)["result"] == 0
        x = graphs.c3_input(2, 20 * 512)
        outs.append(np.concatenate([rt.process(x[:, k * 512:k * 512 + (512 if k % 4 else 200)], len(roots), 512 if k % 4 else 200) for k in range(20)], axis=1))
    assert float(np.abs(outs[1][:5]).max()) > 0.05 and float(np.abs(outs[1][6]).max()) == 0.0
    assert float(np.abs(outs[0].astype(np.float64) - outs[1]).max()) <= TOL


def _blocks(rt, x, k0, nb, ch):
    """blocks [k0, k0 + nb) of x [ch, frames] through elemhip_process_blocks -> [nb, ch, 512]"""
    import torch
    xin = torch.from_numpy(np.ascontiguousarray(x[:, k0 * 512:(k0 + nb) * 512].reshape(x.shape[0], nb, 512).transpose(1, 0, 2))).cuda()
    out = torch.empty((nb, ch, 512), dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    rt.process_blocks(nb, ch, out_ptr=out.data_ptr(), in_ptr=xin.data_ptr(), num_inputs=x.shape[0])
    return out.cpu().numpy()


def test_c3_multi_block_launches_vs_restatement(gpu_required):
    """BASELINE configs[2] through the multi-block convolve kernels: 8 channels x 96 000-tap IRs, 130 blocks = two
    launch sets of 64 and a ragged one of 2 (state carried from set to set: spectra ring, overlap, block counter)."""
    rt, x = _c3(hip, 8, 130)
    got = _blocks(rt, x, 0, 130, 8)
    assert rt.stats()["batch_launches"] >= 2      # (the first blocks, while the roots fade in, go block by block)
    ref_rt, _ = _c3(lambda sr, bs: oracle.PortRuntime(sr, bs), 8, 130)
    ref = np.stack([ref_rt.process(x[:, k * 512:(k + 1) * 512], 8, 512) for k in range(130)])
    assert float(np.abs(ref).max()) > 0.2
    assert float(np.abs(got.astype(np.float64) - ref).max()) <= TOL


@pytest.mark.parametrize("ch,batch,blocks", [(8, 256, 600), (2, 1024, 1024 + 1024 + 90)])
def test_c3_large_launch_sets_vs_restatement(gpu_required, ch, batch, blocks):
    """Launch sets longer than the IR has partitions (188): every spectrum a late block needs comes from the same set, and
    the set leaves only its last 188 spectra in the ring. The geometry benchmarks/bench_configs.py c3 times."""
    rt, x = _c3(hip, ch, blocks)
    rt.set_option("batch_blocks", batch)
    got = np.concatenate([_blocks(rt, x, k0, min(batch + 37, blocks - k0), ch) for k0 in range(0, blocks, batch + 37)])
    assert rt.stats()["batch_launches"] >= 2
    plan = rt.describe_plan()
    # r05: the whole-batch sets take the long-partition kernels, reading the caller's input and writing the caller's output in place
    # (a plan of convolvers only: no arena copy, no bus-sum epilogue); the 37-block remainders take the 512-partition kernels
    assert plan["conv_long_sets"] >= 1 and plan["conv_direct_io_sets"] >= 1, (plan["conv_long_sets"], plan["conv_direct_io_sets"])
    ref_rt, _ = _c3(lambda sr, bs: oracle.PortRuntime(sr, bs), ch, blocks)
    ref = np.stack([ref_rt.process(x[:, k * 512:(k + 1) * 512], ch, 512) for k in range(blocks)])
    err = np.abs(got.astype(np.float64) - ref).max(axis=(1, 2))
    assert float(err.max()) <= TOL, f"block {int(err.argmax())}: {err.max():.3e}"


@pytest.mark.parametrize("mode", [0, 1])
def test_partition_mac_on_matrix_cores_and_on_vector_fmas(gpu_required, mode):
    """The launch sets' partition sums (conv.hip elemhip_convolve_batch_mac) in both forms — `conv_mfma` = 1 (default):
    v_mfma_f32_4x4x1_16B_f32 Toeplitz tiles; 0: packed vector FMAs — each against the restatement (itself pinned to the wasm
    recordings), 4 channels x 96 000-tap IRs, a 256-block set, a ragged 93-block one and a 7-block one (fewer blocks than
    a wave's 16-block tile run; partitions split over two workgroups)."""
    ch, blocks = 4, 256 + 93 + 7
    rt, x = _c3(hip, ch, blocks)
    rt.set_option("batch_blocks", 256)
    rt.set_option("conv_mfma", mode)
    rt.set_option("conv_long", 0)        # (r05: a 256-block set of these IRs would take the long-partition kernels, not this MAC)
    got = np.concatenate([_blocks(rt, x, 0, 256, ch), _blocks(rt, x, 256, 93, ch), _blocks(rt, x, 349, 7, ch)])
    assert rt.stats()["batch_launches"] >= 3
    ref_rt, _ = _c3(lambda sr, bs: oracle.PortRuntime(sr, bs), ch, blocks)
    ref = np.stack([ref_rt.process(x[:, k * 512:(k + 1) * 512], ch, 512) for k in range(blocks)])
    assert float(np.abs(ref).max()) > 0.2
    err = np.abs(got.astype(np.float64) - ref).max(axis=(1, 2))
    assert float(err.max()) <= TOL, f"block {int(err.argmax())}: {err.max():.3e}"


@pytest.mark.parametrize("taps", [300, 700, 3000, 40000])
def test_multi_block_and_single_block_calls_interleave(gpu_required, taps):
    """One stream rendered by alternating elemhip_process_blocks (multi-block kernels) and elemhip_process (main +
    helper workgroups): partitions 1, 2, 6 (fewer than the blocks of a launch set) and 79; a new IR half way restarts
    the convolver from silence in both engines."""
    from elementary_amd import el
    ir_a, ir_b = graphs.c3_impulse_response(0, taps), graphs.c3_impulse_response(1, max(taps // 2, 5))
    roots = [el.convolve({"path": "ir", "key": "a"}, el.in_({"channel": 0})),
             el.mul(0.5, el.convolve({"path": "ir", "key": "b"}, el.mul(0.7, el.in_({"channel": 1}))))]
    plan = [("blocks", 10), ("one", 3), ("blocks", 70), ("one", 1), ("swap", 0), ("blocks", 66), ("one", 2), ("blocks", 5)]
    total = sum(n for kind, n in plan if kind != "swap")
    x = graphs.c3_input(2, total * 512)
    a, c = hip(48000.0, 512), oracle.PortRuntime(48000.0, 512)
    for rt in (a, c):
        assert rt.add_shared_resource("ir", ir_a)
        assert rt.render(*roots)["result"] == 0
    k, batched = 0, 0
    for kind, n in plan:
        if kind == "swap":
            for rt in (a, c):
                assert rt.add_shared_resource("ir2", ir_b)
                assert rt.render(el.convolve({"path": "ir2", "key": "a"}, el.in_({"channel": 0})), roots[1])["result"] == 0
            continue
        ref = np.stack([c.process(x[:, (k + i) * 512:(k + i + 1) * 512], 2, 512) for i in range(n)])
        if kind == "blocks":
            got = _blocks(a, x, k, n, 2)
            batched += 1
        else:
            got = np.stack([a.process(x[:, (k + i) * 512:(k + i + 1) * 512], 2, 512) for i in range(n)])
        assert float(np.abs(got.astype(np.float64) - ref).max()) <= TOL, (kind, n, k)
        k += n
    assert a.stats()["batch_launches"] >= batched     # the multi-block kernels did run


@pytest.mark.parametrize("taps", [16384, 40000, 96000])
def test_long_partition_sets_interleave_with_every_other_path(gpu_required, taps):
    """conv_long.inc: launch sets of a multiple of 8 blocks render IRs of >= 32 partitions with 4096-sample partitions (8192-point
    overlap-save, the history in front of a set from the node's time-domain input ring). One stream through sets of 8, 64, 72 and
    16 blocks (long partitions), 10, 5 and 3 blocks (512-sample partitions) and single-block calls (main + helper workgroups), a
    new IR half way (the convolver restarts from silence), two nodes with different IR lengths in one level — each stretch against
    the restatement; then the same stream with `conv_long` = 0: the two evaluations agree far inside the tolerance."""
    from elementary_amd import el
    ir_a, ir_b, ir_c = graphs.c3_impulse_response(0, taps), graphs.c3_impulse_response(1, 20000), graphs.c3_impulse_response(2, 3000)
    roots = [el.convolve({"path": "ir", "key": "a"}, el.in_({"channel": 0})),
             el.mul(0.5, el.convolve({"path": "irc", "key": "c"}, el.mul(0.7, el.in_({"channel": 1}))))]      # 6 partitions: never long
    plan = [("blocks", 8), ("one", 2), ("blocks", 64), ("blocks", 10), ("blocks", 72), ("one", 1), ("swap", 0), ("blocks", 16), ("blocks", 5),
            ("blocks", 64), ("one", 3), ("blocks", 3), ("blocks", 8)]
    total = sum(n for kind, n in plan if kind != "swap")
    x = graphs.c3_input(2, total * 512)
    outs = {}
    for long_on in (1, 0):
        a, c = hip(48000.0, 512), oracle.PortRuntime(48000.0, 512)
        a.set_option("conv_long", long_on)
        for rt in (a, c):
            assert rt.add_shared_resource("ir", ir_a) and rt.add_shared_resource("irc", ir_c)
            assert rt.render(*roots)["result"] == 0
        k, got_all = 0, []
        for kind, n in plan:
            if kind == "swap":
                for rt in (a, c):
                    assert rt.add_shared_resource("ir2", ir_b)
                    assert rt.render(el.convolve({"path": "ir2", "key": "a"}, el.in_({"channel": 0})), roots[1])["result"] == 0
                continue
            ref = np.stack([c.process(x[:, (k + i) * 512:(k + i + 1) * 512], 2, 512) for i in range(n)])
            if kind == "blocks":
                got = _blocks(a, x, k, n, 2)
            else:
                got = np.stack([a.process(x[:, (k + i) * 512:(k + i + 1) * 512], 2, 512) for i in range(n)])
            err = np.abs(got.astype(np.float64) - ref).max(axis=(1, 2))
            assert float(err.max()) <= TOL, (long_on, kind, n, k, int(err.argmax()), float(err.max()))
            got_all.append(got)
            k += n
        outs[long_on] = np.concatenate(got_all)
        assert a.stats()["batch_launches"] >= 8
        if long_on:
            assert a.describe_plan()["conv_long_sets"] >= 4, a.describe_plan()["conv_long_sets"]     # the long-partition kernels did run
        else:
            assert a.describe_plan()["conv_long_sets"] == 0
    assert float(np.abs(outs[1].astype(np.float64) - outs[0]).max()) <= 5e-7


def test_a_convolver_that_sits_out_a_plan_is_repaired_when_it_comes_back(gpu_required):
    """ADVICE r05 (engine.cpp, one engine-wide `convOverlapStale` flag): a long-partition launch set leaves a convolver's `overlap`
    and 512-partition spectra ring to be rebuilt on demand. The node may then sit out a plan in which OTHER convolvers render
    block-at-a-time (their repair must not clear X's debt) and come back later onto the block-at-a-time path: staleness is kept
    per node. Every stretch against the restatement driven through the same commits (a fading-out root keeps its convolver
    running for the blocks of the fade, GraphRenderSequence.h:239-262)."""
    from elementary_amd import el
    ir_x, ir_y, ir_w = graphs.c3_impulse_response(0, 20000), graphs.c3_impulse_response(1, 18000), graphs.c3_impulse_response(2, 3000)
    X = el.convolve({"path": "irx", "key": "x"}, el.in_({"channel": 0}))
    Y = el.convolve({"path": "iry", "key": "y"}, el.in_({"channel": 0}))
    W = el.convolve({"path": "irw", "key": "w"}, el.in_({"channel": 1}))
    steps = [("render", (X,)), ("blocks", 16),              # X through a long-partition set: stale
             ("render", (Y,)), ("blocks", 16),              # X fades out under Y's plan (another long set), then stops running
             ("render", (Y, W)), ("one", 3),                # a plan WITHOUT X, block at a time: Y is repaired — X's debt must survive
             ("render", (X, W)), ("one", 6),                # X back, block at a time: its overlap has to be rebuilt now
             ("blocks", 8), ("one", 2)]
    total = sum(n for kind, n in steps if kind != "render")
    x = graphs.c3_input(2, total * 512)
    a, c = hip(48000.0, 512), oracle.PortRuntime(48000.0, 512)
    for rt in (a, c):
        assert rt.add_shared_resource("irx", ir_x) and rt.add_shared_resource("iry", ir_y) and rt.add_shared_resource("irw", ir_w)
    k = 0
    for kind, arg in steps:
        if kind == "render":
            for rt in (a, c):
                assert rt.render(*arg)["result"] == 0
            continue
        n = arg
        ref = np.stack([c.process(x[:, (k + i) * 512:(k + i + 1) * 512], 2, 512) for i in range(n)])
        got = _blocks(a, x, k, n, 2) if kind == "blocks" else np.stack([a.process(x[:, (k + i) * 512:(k + i + 1) * 512], 2, 512) for i in range(n)])
        err = np.abs(got.astype(np.float64) - ref).max(axis=(1, 2))
        assert float(err.max()) <= TOL, (kind, n, k, int(err.argmax()), float(err.max()))
        k += n
    assert a.describe_plan()["conv_long_sets"] >= 3


def test_long_partition_sets_back_to_back_carry_their_spectra(gpu_required):
    """r06: the long-partition spectra `U` are a ring carried from set to set — a set's Q - 1 history rows are the previous set's last
    rows, not transformed again — valid only while nothing else moved the node's block counter and the convolver state is the same one
    (conv_long.inc, scratch header). Back-to-back sets of changing sizes (the ring wraps at different places), two nodes with IRs of
    different length in one level (different history depths), a single block in between (the carry is off for one set), an IR swap
    (a new state: nothing may be carried), and two engines whose scratch geometry differs (`batch_blocks` 64 and 256): every
    stretch against the restatement."""
    from elementary_amd import el
    ir_a, ir_b, ir_c = graphs.c3_impulse_response(0, 96000), graphs.c3_impulse_response(1, 20000), graphs.c3_impulse_response(2, 40000)
    roots = [el.convolve({"path": "ira", "key": "a"}, el.in_({"channel": 0})), el.convolve({"path": "irb", "key": "b"}, el.in_({"channel": 1}))]
    steps = [("blocks", 8), ("blocks", 16), ("blocks", 8), ("blocks", 64), ("blocks", 24), ("blocks", 8), ("one", 1), ("blocks", 16), ("blocks", 16),
             ("swap", 0), ("blocks", 8), ("blocks", 32), ("blocks", 8), ("blocks", 64)]
    total = sum(n for kind, n in steps if kind != "swap")
    x = graphs.c3_input(2, total * 512)
    for batch in (64, 256):
        a, c = hip(48000.0, 512), oracle.PortRuntime(48000.0, 512)
        a.set_option("batch_blocks", batch)
        for rt in (a, c):
            assert rt.add_shared_resource("ira", ir_a) and rt.add_shared_resource("irb", ir_b) and rt.add_shared_resource("irc", ir_c)
            assert rt.render(*roots)["result"] == 0
        k = 0
        for kind, n in steps:
            if kind == "swap":
                for rt in (a, c):
                    assert rt.render(el.convolve({"path": "irc", "key": "a"}, el.in_({"channel": 0})), roots[1])["result"] == 0
                continue
            ref = np.stack([c.process(x[:, (k + i) * 512:(k + i + 1) * 512], 2, 512) for i in range(n)])
            got = _blocks(a, x, k, n, 2) if kind == "blocks" else np.stack([a.process(x[:, (k + i) * 512:(k + i + 1) * 512], 2, 512) for i in range(n)])
            err = np.abs(got.astype(np.float64) - ref).max(axis=(1, 2))
            assert float(err.max()) <= TOL, (batch, kind, n, k, int(err.argmax()), float(err.max()))
            k += n
        assert a.describe_plan()["conv_long_sets"] >= 10


@pytest.mark.parametrize("seed", [1, 2, 3, 4])
def test_random_call_sequences_through_every_convolve_path(gpu_required, seed):
    """Seeded random streams over the convolver's three evaluations — single blocks, 512-partition sets (any size), long-partition
    sets (multiples of 8, spectra carried between consecutive ones) — for 1-3 nodes with IRs of random lengths (some below the
    32-partition threshold of the long path), random `batch_blocks`, an IR swap at a random place: every stretch against the
    restatement. (The hand-written interleaving tests fix the orders somebody thought of.)"""
    from elementary_amd import el
    rng = np.random.default_rng(1000 + seed)
    n_nodes = int(rng.integers(1, 4))
    lens = [int(rng.choice([3000, 17000, 20000, 33000, 50000, 96000])) for _ in range(n_nodes)]
    irs = [graphs.c3_impulse_response(i, lens[i]) for i in range(n_nodes)]
    roots = [el.convolve({"path": f"ir{i}", "key": f"n{i}"}, el.in_({"channel": i})) for i in range(n_nodes)]
    batch = int(rng.choice([32, 64, 128]))
    steps = []
    for _ in range(14):
        kind = rng.choice(["long", "long", "long", "ragged", "one"])
        if kind == "long":
            steps.append(("blocks", int(8 * rng.integers(1, batch // 8 + 1))))
        elif kind == "ragged":
            steps.append(("blocks", int(rng.integers(2, 20))))
        else:
            steps.append(("one", int(rng.integers(1, 4))))
    swap_at = int(rng.integers(3, 11))
    total = sum(n for _, n in steps)
    x = graphs.c3_input(n_nodes, total * 512)
    a, c = hip(48000.0, 512), oracle.PortRuntime(48000.0, 512)
    a.set_option("batch_blocks", batch)
    for rt in (a, c):
        for i, ir in enumerate(irs):
            assert rt.add_shared_resource(f"ir{i}", ir)
        assert rt.add_shared_resource("swap", graphs.c3_impulse_response(7, 24000))
        assert rt.render(*roots)["result"] == 0
    k = 0
    for idx, (kind, n) in enumerate(steps):
        if idx == swap_at:
            new_roots = [el.convolve({"path": "swap", "key": "n0"}, el.in_({"channel": 0}))] + roots[1:]
            for rt in (a, c):
                assert rt.render(*new_roots)["result"] == 0
        ref = np.stack([c.process(x[:, (k + i) * 512:(k + i + 1) * 512], n_nodes, 512) for i in range(n)])
        got = _blocks(a, x, k, n, n_nodes) if kind == "blocks" else np.stack([a.process(x[:, (k + i) * 512:(k + i + 1) * 512], n_nodes, 512) for i in range(n)])
        err = np.abs(got.astype(np.float64) - ref).max(axis=(1, 2))
        assert float(err.max()) <= TOL, (seed, lens, batch, idx, kind, n, k, int(err.argmax()), float(err.max()))
        k += n


def test_long_mac_variants_agree(gpu_required):
    """Option `conv_long_mac_lds`: the long-partition sums in their four forms — 0 the register kernel (ships), 1 the LDS-tiled kernel
    (64 bins x 32 chunks per workgroup, rows shared through LDS), 2 runs of 32 chunks per thread, 3 odd runs walking the taps
    oldest-first. Forms 0, 1 and 2 add every (bin, chunk)'s taps in the same order: identical bits; form 3 reorders every second
    run: within 5e-7. Each against the restatement; sets of 64, 24 and 8 blocks (ragged last runs), two IR lengths."""
    from elementary_amd import el
    irs = [graphs.c3_impulse_response(0, 96000), graphs.c3_impulse_response(1, 30000)]
    roots = [el.convolve({"path": f"ir{i}", "key": f"n{i}"}, el.in_({"channel": i})) for i in range(2)]
    sets = [8, 64, 24, 64, 8, 40]
    x = graphs.c3_input(2, sum(sets) * 512)
    c = oracle.PortRuntime(48000.0, 512)
    for i, ir in enumerate(irs):
        assert c.add_shared_resource(f"ir{i}", ir)
    assert c.render(*roots)["result"] == 0
    ref = np.stack([c.process(x[:, k * 512:(k + 1) * 512], 2, 512) for k in range(sum(sets))])
    outs = {}
    for mode in (0, 1, 2, 3):
        a = hip(48000.0, 512)
        a.set_option("batch_blocks", 64)
        a.set_option("conv_long_mac_lds", mode)
        for i, ir in enumerate(irs):
            assert a.add_shared_resource(f"ir{i}", ir)
        assert a.render(*roots)["result"] == 0
        k, got = 0, []
        for n in sets:
            got.append(_blocks(a, x, k, n, 2))
            k += n
        outs[mode] = np.concatenate(got)
        assert float(np.abs(outs[mode].astype(np.float64) - ref).max()) <= TOL, mode
        assert a.describe_plan()["conv_long_sets"] >= 4
    assert np.array_equal(outs[1], outs[0]) and np.array_equal(outs[2], outs[0])
    assert float(np.abs(outs[3].astype(np.float64) - outs[0]).max()) <= 5e-7


def test_the_sample_clock_after_direct_io_sets(gpu_required):
    """A plan of long-partition convolvers only renders its launch sets with direct I/O: no epilogue kernel advances the device's
    sample clock, and since r06 no parameter patch per set either — the clock is caught up when something is about to read it.
    Convolver sets, then a graph whose samples ARE the clock (`time`, `metro`: wasm/SampleTime.h, wasm/Metro.h) block at a time,
    through a launch set, and through the opt-in resident kernel: every block against the restatement."""
    from elementary_amd import el
    ir = graphs.c3_impulse_response(0, 20000)
    conv = [el.convolve({"path": "ir", "key": "x"}, el.in_({"channel": 0}))]
    clock = [el.mul(1e-4, el.time()), el.metro({"interval": 3.0})]
    for resident in (0, 1):
        a, c = hip(48000.0, 512), oracle.PortRuntime(48000.0, 512)
        a.set_option("resident", resident)
        for rt in (a, c):
            assert rt.add_shared_resource("ir", ir)
        x = graphs.c3_input(1, 200 * 512)
        k = 0
        plan = [("render", conv), ("blocks", 16), ("blocks", 64), ("render", clock), ("one", 12), ("blocks", 16), ("one", 4),
                ("render", conv), ("blocks", 24), ("render", clock), ("one", 6)]
        for kind, arg in plan:
            if kind == "render":
                for rt in (a, c):
                    assert rt.render(*arg)["result"] == 0
                n_out = len(arg)
                continue
            ref = np.stack([c.process(x[:, (k + i) * 512:(k + i + 1) * 512], n_out, 512) for i in range(arg)])
            if kind == "blocks":
                got = _blocks(a, x, k, arg, n_out)
            else:
                got = np.stack([a.process(x[:, (k + i) * 512:(k + i + 1) * 512], n_out, 512) for i in range(arg)])
            scale = max(1.0, float(np.abs(ref).max()))
            err = np.abs(got.astype(np.float64) - ref).max(axis=(1, 2))
            assert float(err.max()) <= TOL * scale, (resident, kind, arg, k, int(err.argmax()), float(err.max()))
            k += arg
        assert a.describe_plan()["conv_direct_io_sets"] >= 2


def test_partial_blocks_switch_the_multi_block_path_off(gpu_required):
    """After a call of fewer than 512 frames a convolver's input block may be partly filled at a call boundary: the
    engine goes back to block-at-a-time launches for plans with convolvers (still the same samples)."""
    from elementary_amd import el
    ir = graphs.c3_impulse_response(0, 5000)
    x = graphs.c3_input(1, 40 * 512)
    a, c = hip(48000.0, 512), oracle.PortRuntime(48000.0, 512)
    for rt in (a, c):
        assert rt.add_shared_resource("ir", ir)
        assert rt.render(el.convolve({"path": "ir"}, el.in_({"channel": 0})))["result"] == 0
    got = [a.process(x[:, :200], 1, 200)]
    ref = [c.process(x[:, :200], 1, 200)]
    pos = 200
    before = a.stats()["batch_launches"]
    xs = np.concatenate([x[:, pos:], np.zeros((1, 512), np.float32)], axis=1)
    got.append(_blocks(a, xs, 0, 30, 1).transpose(1, 0, 2).reshape(1, -1))
    assert a.stats()["batch_launches"] == before
    ref.append(np.concatenate([c.process(xs[:, i * 512:(i + 1) * 512], 1, 512) for i in range(30)], axis=1))
    assert float(np.abs(np.concatenate(got, axis=1).astype(np.float64) - np.concatenate(ref, axis=1)).max()) <= TOL


@pytest.mark.parametrize("batch", [1, 64, 256])
def test_c3_two_channels_vs_the_wasm_recording(gpu_required, batch):
    """Two channels of BASELINE configs[2] against outputs RECORDED FROM THE REFERENCE'S WASM ENGINE
    (tests/golden/convolve_wasm_c3x2.f32, 200 blocks: past the 188 partitions of the IR), block at a time (batch = 1) and
    through 64- / 256-block launch sets — not against the restatement."""
    import json
    import os
    here = os.path.dirname(os.path.abspath(__file__))
    man = json.load(open(os.path.join(here, "golden", "convolve_wasm_c3x2.json")))
    gold = np.fromfile(os.path.join(here, "golden", "convolve_wasm_c3x2.f32"), dtype="<f4").reshape(man["channels"], -1)
    ch, nb = man["channels"], man["blocks"]
    rt, x = _c3(hip, ch, nb)
    rt.set_option("batch_blocks", batch)
    if batch == 1:
        got = np.concatenate([rt.process(x[:, k * 512:(k + 1) * 512], ch, 512) for k in range(nb)], axis=1)
    else:
        got = _blocks(rt, x, 0, nb, ch).transpose(1, 0, 2).reshape(ch, nb * 512)
        assert rt.stats()["batch_launches"] >= 1
    err = np.abs(got.astype(np.float64) - gold)
    assert float(err.max()) <= TOL, f"frame {int(err.argmax())}: {err.max():.3e}"
