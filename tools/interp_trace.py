"""Task-by-task timeline of ONE block of an island through the interpreter island kernel (island.inc trace hook: per wave
{opcode | stage << 16, start, end}): where a single block's latency goes. Usage: python tools/interp_trace.py [c1|c2] [level]"""
import os as _os, sys as _sys; _R = _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))); _sys.path[:0] = [_R, _os.path.join(_R, 'tests')]
import ctypes as C
import re
import sys

from elementary_amd import graphs
from elementary_amd.runtime import Runtime, load_library

lib = load_library()
lib.elemhip_trace_level.argtypes = [C.c_void_p, C.c_size_t, C.c_uint32, C.POINTER(C.c_ulonglong), C.c_size_t]
which = sys.argv[1] if len(sys.argv) > 1 else "c1"
level = int(sys.argv[2]) if len(sys.argv) > 2 else 0
src = open(_os.path.join(_R, "elementary_amd", "csrc", "device.h")).read()
body = src[src.index("enum Op"):]
body = body[body.index("{") + 1:body.index("OP_COUNT_")]
names = [m for m in re.findall(r"\bOP_[A-Z0-9_]+\b", re.sub(r"//.*", "", body))]
if which == "c2":
    sr, roots, nout = graphs.C2_SAMPLE_RATE, graphs.c2_graph(voices=256), 2
else:
    sr, roots, nout = graphs.C1_SAMPLE_RATE, graphs.c1_graph(), 2
rt = Runtime(sr, 512, device=0)
rt.set_option("specialize", 0)
assert rt.render(*roots)["result"] == 0
for _ in range(20):
    rt.process(None, nout, 512)
buf = (C.c_ulonglong * (8 * 192))()
for _ in range(3):
    rc = lib.elemhip_trace_level(rt._h, nout, level, buf, 8 * 192)
    assert rc == 0, rc
print(rt.stats())
t0 = min(buf[w * 192 + 1] for w in range(8) if buf[w * 192 + 1])
for w in range(8):
    o = w * 192
    k, ts, tp, te = buf[o], buf[o + 1], buf[o + 2], buf[o + 3]
    if not ts:
        continue
    print(f"wave{w}: start {ts - t0} prologue done {tp - t0} end {te - t0}  ({k} tasks)")
    for j in range(min(int(k), 62)):
        d, a, b = buf[o + 3 * (j + 2)], buf[o + 3 * (j + 2) + 1], buf[o + 3 * (j + 2) + 2]
        op, stage = d & 0xFFFF, (d >> 16) & 0xFF
        print(f"    stage {stage:2d} {names[op] if op < len(names) else op:16s} {a - t0:7d} .. {b - t0:7d}  ({b - a:6d})")
