"""Lane-packing of isomorphic islands (plan.cpp step 2b): K same-shape islands of a launch level merged into one workgroup,
their float recurrences sharing a wavefront lane by lane. Packing only changes where a node is rendered, never a sample:
every packed engine must equal the unpacked engine BIT FOR BIT, and the reference engine within 1e-6."""
import numpy as np
import pytest

from elementary_amd import graphs

pytestmark = pytest.mark.gpu
TOL = 1e-6


def _checker(sr, bs):
    import oracle
    return oracle.RefRuntime(sr, bs) if oracle.have_ref() else oracle.PortRuntime(sr, bs)


def _hip(sr, **opts):
    from elementary_amd.runtime import Runtime
    rt = Runtime(sr, 512, device=0)
    for k, v in opts.items():
        rt.set_option(k, v)
    return rt


def _planar(rt, n_out, blocks):
    return rt.process_blocks_host(None, n_out, blocks * 512)


@pytest.mark.parametrize("spec", [0, 2])
@pytest.mark.parametrize("k", [2, 3, 4])
def test_packed_voices_equal_unpacked(gpu_required, spec, k):
    """48 C2 voices, K voices per island when forced, both kernel families, 150 blocks through launch sets of 32. Packed islands
    evaluate the svf coefficients inside the scan (`fuse_svf_coef` = 2: no 6-slot scratch per voice), which is what leaves
    K = 4 two buffer sets; the unpacked engine keeps the separate pre-pass — the two forms are bit-identical too."""
    a = _hip(graphs.C2_SAMPLE_RATE, specialize=spec, batch_blocks=32, pack_islands=k)
    b = _hip(graphs.C2_SAMPLE_RATE, specialize=spec, batch_blocks=32, pack_islands=1)
    c = _checker(graphs.C2_SAMPLE_RATE, 512)
    roots = graphs.c2_graph(voices=48)
    for rt in (a, b, c):
        assert rt.render(*roots)["result"] == 0
    pa, pb = a.describe_plan(), b.describe_plan()
    assert pb["pack_k"] == 1 and pb["level_sizes"][0] == 48
    assert pa["pack_k"] == k and pa["level_sizes"][0] == -(-48 // k)
    assert pa["islands"][0]["copies"] >= 2
    got, ref_hip = _planar(a, 2, 150), _planar(b, 2, 150)
    assert np.array_equal(got, ref_hip)
    ref = np.concatenate([c.process(None, 2, 512) for _ in range(150)], axis=1)
    assert float(np.abs(got - ref).max()) <= TOL * max(1.0, float(np.abs(ref).max()))
    if spec:
        st = a.stats()
        assert st["spec_launches"] > 0 and all(a.spec_info(q)["state"] == 1 for q in range(st["spec_shapes"])), st


@pytest.mark.parametrize("spec", [0, 2])
def test_fused_svf_coefficients_equal_the_pre_pass(gpu_required, spec):
    """`fuse_svf_coef` = 1 on unpacked voices vs 0: the same double operations in another place — bit-identical; modulated cutoff
    (C2 voices), constant cutoff and resonance, and a shelf next to it (never fused)."""
    from elementary_amd import el
    x = el.in_({"channel": 0})
    roots = graphs.c2_graph(voices=6) + [el.lowpass(900.0, 1.1, x), el.highpass(el.add(1000.0, el.mul(500.0, x)), 0.8, x),
                                         el.bandpass(700.0, el.add(2.0, x), x), el.lowshelf(300.0, 0.9, 3.0, x)]
    a = _hip(graphs.C2_SAMPLE_RATE, specialize=spec, batch_blocks=16, fuse_svf_coef=1)
    b = _hip(graphs.C2_SAMPLE_RATE, specialize=spec, batch_blocks=16, fuse_svf_coef=0)
    c = _checker(graphs.C2_SAMPLE_RATE, 512)
    for rt in (a, b, c):
        assert rt.render(*roots)["result"] == 0
    from helpers import lcg_noise
    xin = np.stack([lcg_noise(60 * 512, 21, 0.5)])
    ya, yb = a.process_blocks_host(xin, len(roots), 60 * 512), b.process_blocks_host(xin, len(roots), 60 * 512)
    assert np.array_equal(ya, yb)
    ref = np.concatenate([c.process(xin[:, k * 512:(k + 1) * 512], len(roots), 512) for k in range(60)], axis=1)
    assert float(np.abs(ya - ref).max()) <= TOL * max(1.0, float(np.abs(ref).max()))


def test_auto_packing_follows_the_cu_count(gpu_required):
    """Auto mode packs when a launch level has more stateful islands than the device has CUs: with the CU count it plans for
    set to 16 and up to 3 per island allowed, 40 voices become 14 islands of 3 (ceil(40 / 16) = 3), 16 voices stay one per island."""
    a, b = _hip(graphs.C2_SAMPLE_RATE, cu_count=16, pack_max=3, batch_blocks=16), _hip(graphs.C2_SAMPLE_RATE, cu_count=16, pack_max=3, batch_blocks=16)
    c = _checker(graphs.C2_SAMPLE_RATE, 512)
    assert a.render(*graphs.c2_graph(voices=40))["result"] == 0 and c.render(*graphs.c2_graph(voices=40))["result"] == 0
    assert b.render(*graphs.c2_graph(voices=16))["result"] == 0
    assert a.describe_plan()["pack_k"] == 3 and a.describe_plan()["level_sizes"][0] == 14
    assert b.describe_plan()["pack_k"] == 1 and b.describe_plan()["level_sizes"][0] == 16
    got = _planar(a, 2, 70)
    ref = np.concatenate([c.process(None, 2, 512) for _ in range(70)], axis=1)
    assert float(np.abs(got - ref).max()) <= TOL * max(1.0, float(np.abs(ref).max()))


def test_render_jobs_with_a_root_each_are_not_packed(gpu_required):
    """Islands are only merged within one root sequence (an island renders while ITS root runs): C4 render jobs, a root per
    job, stay one per workgroup whatever the option says — and render as before."""
    a = _hip(graphs.C4_SAMPLE_RATE, batch_blocks=16, pack_islands=4)
    c = _checker(graphs.C4_SAMPLE_RATE, 512)
    roots = [graphs.c4_instance(k) for k in range(10)]
    for rt in (a, c):
        assert rt.render(*roots)["result"] == 0
    p = a.describe_plan()
    assert p["pack_k"] == 1 and p["level_sizes"][0] == 10
    got = _planar(a, 10, 40)
    ref = np.concatenate([c.process(None, 10, 512) for _ in range(40)], axis=1)
    assert float(np.abs(got - ref).max()) <= TOL


@pytest.mark.parametrize("spec", [0, 2])
@pytest.mark.parametrize("k", [2, 4])
def test_render_jobs_packed_across_roots(gpu_required, spec, k):
    """`pack_roots` = 1 (opt-in): islands of DIFFERENT root sequences share a workgroup when all of those roots are active at plan
    time — 12 C4 render jobs, a root each, K per island. Bit-identical to the unpacked engine, within 1e-6 of the reference; a job
    whose root is switched off is re-planned (the commit that changes the targets) and rendered like the reference; a call that
    asks for fewer outputs than the packed roots' channels is refused."""
    a = _hip(graphs.C4_SAMPLE_RATE, specialize=spec, batch_blocks=16, pack_islands=k, pack_roots=1)
    b = _hip(graphs.C4_SAMPLE_RATE, specialize=spec, batch_blocks=16, pack_islands=1)
    c = _checker(graphs.C4_SAMPLE_RATE, 512)
    roots = [graphs.c4_instance(i) for i in range(12)]
    for rt in (a, b, c):
        assert rt.render(*roots)["result"] == 0
    pa = a.describe_plan()
    assert pa["pack_k"] == k and pa["level_sizes"][0] == 2 * -(-6 // k), pa["level_sizes"]      # two shapes (odd / even jobs) of 6 islands each
    got, ref_hip = _planar(a, 12, 60), _planar(b, 12, 60)
    assert np.array_equal(got, ref_hip)
    ref = np.concatenate([c.process(None, 12, 512) for _ in range(60)], axis=1)
    assert float(np.abs(got - ref).max()) <= TOL
    if spec:
        assert a.stats()["spec_launches"] > 0
    with pytest.raises(Exception):
        a.process_blocks_host(None, 5, 4 * 512)          # fewer outputs than the packed roots' channels
    # jobs 3 and 8 end: their roots fade out in the new plan (not packed with anyone), the rest stay packed
    sil = [graphs.c4_instance(i) if i not in (3, 8) else 0.0 for i in range(12)]
    for rt in (a, c):
        assert rt.render(*sil)["result"] == 0
    got2 = _planar(a, 12, 40)
    ref2 = np.concatenate([c.process(None, 12, 512) for _ in range(40)], axis=1)
    assert float(np.abs(got2 - ref2).max()) <= TOL
