import os as _os, sys as _sys; _R = _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))); _sys.path[:0] = [_R, _os.path.join(_R, 'tests')]
import sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from elementary_amd import graphs
from elementary_amd.runtime import Runtime
roots = graphs.c2_graph(voices=16)
b = Runtime(48000.0, 512); assert b.render(*roots)["result"] == 0
nb = 37
ref = np.stack([b.process(None, 2, 512) for _ in range(nb)])
for copies in (1, 2, 3, 4):
  for bb in (16, 5):
    a = Runtime(48000.0, 512); a.set_option("pipeline_copies", copies); a.set_option("batch_blocks", bb)
    assert a.render(*roots)["result"] == 0
    out = torch.zeros((nb, 2, 512), dtype=torch.float32, device="cuda")
    a.process_blocks(nb, 2, out_ptr=out.data_ptr())
    o = out.cpu().numpy()
    bad = [k for k in range(nb) if not np.array_equal(o[k], ref[k])]
    print("copies", copies, "batch", bb, "bad blocks", bad, "nan" if np.isnan(o).any() else "")
