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
    """the hipGraph multi-block path renders the same samples as block-at-a-time process()."""
    import torch
    ch, blocks = 2, 24
    rt, x = _c3(hip, ch, blocks)
    ref = np.stack([rt.process(x[:, k * 512:(k + 1) * 512], ch, 512) for k in range(blocks)])
    rt2, _ = _c3(hip, ch, blocks)
    xin = torch.from_numpy(np.ascontiguousarray(x.reshape(ch, blocks, 512).transpose(1, 0, 2))).cuda()
    out = torch.empty((blocks, ch, 512), dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    rt2.process_blocks(blocks, ch, out_ptr=out.data_ptr(), in_ptr=xin.data_ptr(), num_inputs=ch)
    assert float(np.abs(out.cpu().numpy() - ref).max()) <= 1e-7


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
