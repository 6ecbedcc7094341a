"""The C++ façade (include/elemhip/Runtime.hpp) driven the way the reference's hosts drive elem::Runtime<float>
(cli/Benchmark.cpp:31-112): tests/native/facade_host.cpp is compiled against the reference headers (oracle/Makefile
`facade`, into oracle/_ref/) and uses applyInstructions(js::Array), registerNodeType with the reference's own
MetronomeNode / SampleTimeNode classes (call-out nodes), process(), snapshot(), getSharedResourceMapKeys(), gc()."""
import json
import os
import subprocess
import tempfile

import numpy as np
import pytest

from elementary_amd import el
from elementary_amd.reconciler import Renderer, batch_to_json, create_node

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "oracle", "_ref", "facade_host")
LIB = os.path.join(ROOT, "elementary_amd", "libelemhip.so")


def _graph(custom: bool):
    """custom=True: the metronome / sample clock are the host-registered CPU node types; False: the built-in ones."""
    metro = create_node("cpumetro" if custom else "metro", {"interval": 7.0}, [])
    tme = create_node("cputime" if custom else "time", {}, [])
    gate = el.mul(0.5, metro)                                            # call-out node feeding a GPU node
    ramp = el.table({"path": "ramp"}, el.phasor(3.0))
    shaped = el.add(el.mul(1e-5, tme), el.mul(gate, el.cycle(220.0)))   # two call-out nodes, GPU nodes either side
    return [gate, shaped, ramp]


def _batch(custom: bool):
    sent = []
    r = Renderer(lambda b: sent.append(b) or 0)
    r.render(*_graph(custom))
    return sent[0]


def _run(device: int, blocks: int, custom=True, mode="process"):
    with tempfile.TemporaryDirectory() as d:
        bpath, opath = os.path.join(d, "batch.json"), os.path.join(d, "out.f32")
        open(bpath, "w").write(batch_to_json(_batch(custom)))
        res = subprocess.run([HOST, bpath, str(blocks), "3", opath, str(device), "44100", mode], capture_output=True, text=True, timeout=300)
        assert res.returncode == 0, res.stderr
        out = np.fromfile(opath, dtype=np.float32).reshape(blocks, 3, 512)
        return json.loads(res.stdout.strip().splitlines()[-1]), out


needs_host = pytest.mark.skipif(not (os.path.exists(HOST) and os.path.exists(LIB)), reason="oracle/_ref/facade_host not built (needs /root/reference at build time)")


@needs_host
def test_facade_host_logic_on_a_dry_handle():
    info, out = _run(-1, 2)
    assert info["dup"] == 4 and info["dup_builtin"] == 4            # ReturnCode::NodeTypeAlreadyExists (Runtime.h:483)
    assert info["added"] == 1 and info["added_twice"] == 0          # insert-only resources (SharedResource.h:61-63)
    nodes = {c[1] for c in _batch(True) if c[0] == 0}
    assert info["snapshot_nodes"] == len(nodes)
    assert info["resource_keys"] == 1 and info["first_key"] == "ramp"
    assert info["pruned"] == 0                                       # everything is referenced by the active render sequence


@needs_host
@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["process", "blocks"])
def test_facade_renders_with_reference_node_classes_as_callouts(gpu_required, mode):
    """The reference's MetronomeNode / SampleTimeNode run on the CPU between GPU launch levels; the output must equal the
    reference engine rendering the same graph with its natively registered metro / time. mode "blocks": the whole render
    through one Runtime::processBlocks call (elemhip_process_blocks_host, planar host arrays)."""
    import oracle
    info, got = _run(0, 40, mode=mode)
    chk = oracle.RefRuntime(44100.0, 512)
    table = (np.arange(64, dtype=np.float32) / 64.0)[None, :]
    assert chk.add_shared_resource("ramp", table)
    assert chk.render(*_graph(False))["result"] == 0
    ref = np.stack([chk.process(None, 3, 512, sample_time=512 * k) for k in range(40)])
    assert float(np.abs(ref[:, 0]).max()) == 0.5 and np.abs(ref[:, 2]).max() > 0.5
    assert float(np.abs(got - ref).max()) <= 1e-6
    assert info["events"] == 0
