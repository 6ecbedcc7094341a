import os as _os, sys as _sys; _R = _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))); _sys.path[:0] = [_R, _os.path.join(_R, 'tests')]
import sys, ctypes as C, re
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from elementary_amd import el, graphs
from elementary_amd.runtime import Runtime, load_library
lib = load_library()
lib.elemhip_trace_level.argtypes = [C.c_void_p, C.c_size_t, C.c_uint32, C.POINTER(C.c_ulonglong), C.c_size_t]
src = open('elementary_amd/csrc/device.h').read()
body = src[src.index('enum Op : uint16_t {'):src.index('OP_COUNT_')]
names = {i: t[3:].lower() for i, t in enumerate(re.findall(r'OP_[A-Z0-9_]+', body))}
def trace(rt, nout, level=0):
    buf = (C.c_ulonglong * (8*192))()
    for _ in range(3):
        rc = lib.elemhip_trace_level(rt._h, nout, level, buf, 8*192); assert rc == 0, rc
    base = min(buf[w*192+1] for w in range(8) if buf[w*192+1])
    for w in range(8):
        o = w*192
        nt, ts, tp, te = buf[o], buf[o+1], buf[o+2], buf[o+3]
        print(f" wave{w}: start {ts-base} prologue_end {tp-base} end {te-base} tasks={nt}")
        for k in range(min(nt, 62)):
            d0, t0, t1 = buf[o+3*(k+2)], buf[o+3*(k+2)+1], buf[o+3*(k+2)+2]
            print(f"    {names.get(d0 & 0xFFFF, d0 & 0xFFFF):12s} stage {(d0>>16)&0xFF:2d} blk {(d0>>32)&0xFFFF:2d}  [{t0-base:7d} .. {t1-base:7d}]  {t1-t0:6d} clk")
which = sys.argv[1] if len(sys.argv) > 1 else "voice"
if which in ("pipe32", "c2pipe", "c4pipe", "c1pipe"):
    rt = Runtime(48000.0, 512)
    nout = {"pipe32": 8, "c2pipe": 2, "c4pipe": 128, "c1pipe": 2}[which]
    roots = {"pipe32": lambda: [graphs.c2_voice(k) for k in range(8)], "c2pipe": graphs.c2_graph,
             "c4pipe": lambda: [graphs.c4_instance(k) for k in range(128)], "c1pipe": graphs.c1_graph}[which]()
    assert rt.render(*roots)["result"] == 0
    rt.process_blocks(8, nout); rt.set_option("time_batch", 32)
    buf = (C.c_ulonglong * (8*192))()
    level = int(sys.argv[2]) if len(sys.argv) > 2 else 0      # launch level to trace (its first workgroup)
    rt.set_option("specialize", 0)
    for _ in range(3):
        rc = lib.elemhip_trace_level(rt._h, nout, level, buf, 8*192); assert rc == 0, rc
    base = min(buf[w*192+1] for w in range(8) if buf[w*192+1])
    for w in range(8):
        o = w*192; nt = buf[o]
        rows = []
        for k in range(min(nt, 62)):
            d0, t0, t1 = buf[o+3*(k+2)], buf[o+3*(k+2)+1], buf[o+3*(k+2)+2]
            rows.append((names.get(d0 & 0xFFFF, "?"), (d0 >> 16) & 0xFF, (d0 >> 32) & 0xFFFF, t0 - base, t1 - t0, ((d0 >> 48) & 0xFF) * 64, ((d0 >> 56) & 0xFF) * 64))
        print(f"wave{w} tasks={nt} end={buf[o+3]-base}")
        print("   " + " ".join(f"{n[:4]}{st}b{b}@{t0//1000}k+{d//100/10:g}k[{ga}/{gb}]" for n, st, b, t0, d, ga, gb in rows))
elif which == "voicepipe":
    rt = Runtime(48000.0, 512); assert rt.render(*[graphs.c2_voice(k) for k in range(8)])["result"] == 0
    rt.process_blocks(8, 8); rt.set_option("time_batch", 8); trace(rt, 8)
elif which == "voice":
    rt = Runtime(48000.0, 512); assert rt.render(*[graphs.c2_voice(k) for k in range(8)])["result"] == 0; trace(rt, 8)
elif which == "mul16":
    K = lambda k: el.const({"key": f"k{k}", "value": 100.0 + k})
    def chain(k, n):
        x = K(k)
        for i in range(n): x = el.mul(x, 1.0001 + i * 1e-6)
        return x
    rt = Runtime(48000.0, 512); assert rt.render(*[chain(k, 16) for k in range(8)])["result"] == 0; trace(rt, 8)
elif which == "c2":
    rt = Runtime(48000.0, 512); assert rt.render(*graphs.c2_graph())["result"] == 0; trace(rt, 2, 0); trace(rt, 2, 1)
