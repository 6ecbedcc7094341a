"""dev tool: per-wave task lists of the first island of a graph's plan (dry handle, no GPU)."""
import os as _os, sys as _sys; _R = _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))); _sys.path[:0] = [_R, _os.path.join(_R, 'tests')]

import os, re, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from elementary_amd.runtime import Runtime
from elementary_amd import graphs
src = open(os.path.join(os.path.dirname(__file__), '..', 'elementary_amd/csrc/device.h')).read()
body = src[src.index('enum Op : uint16_t {'):src.index('OP_COUNT_')]
names = {i: t[3:].lower() for i, t in enumerate(re.findall(r'OP_[A-Z0-9_]+', body))}
which = sys.argv[1] if len(sys.argv) > 1 else "c2"
rt = Runtime(48000.0, 512, device=-1)
roots = graphs.c2_graph() if which == "c2" else graphs.c1_graph()
assert rt.render(*roots)["result"] == 0
d = rt.describe_plan()
I = d["islands"][0]
for w, ts in enumerate(I["waves"]): print(w, " ".join(f"{names[o]}@{s}" for o, s in ts))
print({k: v for k, v in I.items() if k != "waves"})
