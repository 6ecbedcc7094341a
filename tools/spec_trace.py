"""Per-wave busy time of a specialised island kernel (island_spec.inc trace hook): which (wave, slot) bounds the block
pipeline. Usage: python tools/spec_trace.py [c2|c4|c1] [blocks_per_launch]"""
import os as _os, sys as _sys; _R = _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))); _sys.path[:0] = [_R, _os.path.join(_R, 'tests')]
import ctypes as C
import sys

import torch  # noqa: F401

from elementary_amd import graphs
from elementary_amd.runtime import Runtime, load_library

lib = load_library()
lib.elemhip_trace_level.argtypes = [C.c_void_p, C.c_size_t, C.c_uint32, C.POINTER(C.c_ulonglong), C.c_size_t]
which = sys.argv[1] if len(sys.argv) > 1 else "c2"
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 64
spec = int(sys.argv[3]) if len(sys.argv) > 3 else 2
if which == "c2":
    sr, roots, nout = graphs.C2_SAMPLE_RATE, graphs.c2_graph(voices=256), 2
elif which == "c4":
    sr, roots, nout = graphs.C4_SAMPLE_RATE, [graphs.c4_instance(k) for k in range(128)], 128
else:
    sr, roots, nout = graphs.C1_SAMPLE_RATE, graphs.c1_graph(), 2
rt = Runtime(sr, 512, device=0)
rt.set_option("specialize", spec)
rt.set_option("batch_blocks", batch)
rt.set_option("time_batch", batch)
import os
for kv in os.environ.get("ELEMHIP_TRACE_OPTS", "").split():      # extra engine options: "key=value key=value"
    k, v = kv.split("=", 1)
    rt.set_option(k, float(v))
assert rt.render(*roots)["result"] == 0
out = torch.zeros((batch * 4, nout, 512), dtype=torch.float32, device="cuda")
rt.process_blocks(batch * 4, nout, out_ptr=out.data_ptr())      # settle fades, warm caches
buf = (C.c_ulonglong * (8 * 192))()
for _ in range(2):
    rc = lib.elemhip_trace_level(rt._h, nout, 0, buf, 8 * 192)
    assert rc == 0, rc
print(rt.stats())
t0 = min(buf[w * 192 + 1] for w in range(8) if buf[w * 192 + 1])
for w in range(8):
    o = w * 192
    k, ts, tp, te = buf[o], buf[o + 1], buf[o + 2], buf[o + 3]
    line = f"wave{w}: start {ts - t0:7d} prologue {tp - ts:6d} total {te - ts:9d} ({(te - ts) / batch:8.0f}/block)"
    for j in range(int(k)):
        busy, runs = buf[o + 4 + 2 * j], buf[o + 5 + 2 * j]
        line += f" | slot{j}: busy {busy / max(1, runs):8.0f}/block x{runs}"
    print(line)
