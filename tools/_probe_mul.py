import os as _os, sys as _sys; _R = _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))); _sys.path[:0] = [_R, _os.path.join(_R, 'tests')]
import sys
sys.path.insert(0, '.')
from elementary_amd import el
from elementary_amd.runtime import Runtime
import torch
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
K = lambda k: el.const({"key": f"k{k}", "value": 100.0 + k})
def chain(k, n):
    x = K(k)
    for i in range(n): x = el.mul(x, 1.0001 + i * 1e-6)
    return x
rt = Runtime(48000.0, 512)
rt.set_option("use_graph", 0)
assert rt.render(*[chain(k, n) for k in range(64)])["result"] == 0
out = torch.zeros((64, 64, 512), device="cuda")
for _ in range(4):
    rt.process_blocks(64, 64, out_ptr=out.data_ptr())
print(rt.time_launches(64, 100))
