"""Multi-output (mc.*) nodes on a LIVE engine: a `path` re-pointed while the node renders, and an output channel that a
later plan consumes for the first time while channel 0 is mid-playback. The reference keeps one reader state for all
channels of an mc node (mc/Sample.h:83-150, mc/SampleSeq.h), so the channels stay sample-aligned."""
import numpy as np
import pytest

from elementary_amd import el

pytestmark = pytest.mark.gpu
TOL = 1e-6


def _pair():
    import oracle
    from elementary_amd.runtime import Runtime
    if not oracle.have_ref():
        pytest.skip("needs oracle/_ref (the port oracle has no mc nodes)")
    return Runtime(44100.0, 512, device=0), oracle.RefRuntime(44100.0, 512)


def _resources():
    t = np.arange(4000, dtype=np.float32)
    return {"/m/a": np.stack([np.sin(t * 0.01), np.cos(t * 0.013)]).astype(np.float32),
            "/m/b": np.stack([0.5 * np.cos(t * 0.021), 0.25 * np.sin(t * 0.005) + 0.1]).astype(np.float32)}


@pytest.mark.parametrize("path", ["process", "process_blocks_host"])
def test_mc_path_repointed_on_a_live_node(gpu_required, path):
    a, c = _pair()
    for rt in (a, c):
        for k, v in _resources().items():
            assert rt.add_shared_resource(k, v)
    outs = []
    for rt in (a, c):
        smp, set_smp = rt.renderer.create_ref("mc.sample", {"path": "/m/a", "mode": "loop", "playbackRate": 0.5}, [el.train(3.0)])
        tbl, set_tbl = rt.renderer.create_ref("mc.table", {"path": "/m/a"}, [el.phasor(5.0)])
        from elementary_amd.reconciler import unpack
        roots = unpack(smp, 2) + unpack(tbl, 2)
        assert rt.render(*roots)["result"] == 0
        ys = []

        def run(nb):
            if rt is a and path == "process_blocks_host":
                y = rt.process_blocks_host(None, 4, nb * 512)
                ys.extend(y[:, k * 512:(k + 1) * 512] for k in range(nb))
            else:
                ys.extend(rt.process(None, 4, 512) for _ in range(nb))
        run(9)
        assert set_smp({"path": "/m/b"}) == 0 and set_tbl({"path": "/m/b"}) == 0      # every channel must move to ITS channel of /m/b
        run(9)
        assert set_smp({"path": "/m/a"}) == 0
        run(5)
        outs.append(np.stack(ys))
    got, ref = outs
    assert np.abs(ref[:, 1]).max() > 0.05 and np.abs(ref[:, 3]).max() > 0.05
    assert float(np.abs(got - ref).max()) <= TOL


def test_mc_channel_first_used_by_a_later_plan(gpu_required):
    """Plan 1 renders channel 0 of an mc.sample (and of an mc.table) only; plan 2, 7 blocks later, also takes channel 1: it
    continues from the node's live reader state, aligned with channel 0. (mc.sampleseq is left out: the reference engine
    itself crashes when only channel 0 of a two-channel mc.sampleseq is consumed.)"""
    a, c = _pair()
    for rt in (a, c):
        for k, v in _resources().items():
            assert rt.add_shared_resource(k, v)
    outs = []
    for rt in (a, c):
        def nodes():
            smp = el.mc.sample({"path": "/m/a", "channels": 2, "mode": "loop", "playbackRate": 0.75}, el.train(2.0))
            tbl = el.mc.table({"path": "/m/b", "channels": 2}, el.phasor(7.0))
            return smp, tbl
        smp, tbl = nodes()
        assert rt.render(smp[0], tbl[0])["result"] == 0
        ys = [np.pad(rt.process(None, 2, 512), ((0, 2), (0, 0))) for _ in range(7)]
        smp, tbl = nodes()
        assert rt.render(smp[0], tbl[0], smp[1], tbl[1])["result"] == 0
        ys += [rt.process(None, 4, 512) for _ in range(9)]
        outs.append(np.stack(ys))
    got, ref = outs
    assert np.abs(ref[8:, 2]).max() > 0.05 and np.abs(ref[8:, 3]).max() > 0.01
    assert float(np.abs(got - ref).max()) <= TOL
