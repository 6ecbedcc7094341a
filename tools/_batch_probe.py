import os as _os, sys as _sys; _R = _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))); _sys.path[:0] = [_R, _os.path.join(_R, 'tests')]
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from elementary_amd import graphs
from elementary_amd.runtime import Runtime
for copies in (3, 4, 5):
    rt = Runtime(48000.0, 512, device=0)
    rt.set_option("pipeline_copies", copies)
    assert rt.render(*graphs.c2_graph())["result"] == 0
    rt.process_blocks(64, 2)
    for batch in (16, 32, 64):
        rt.set_option("batch_blocks", batch); rt.set_option("time_batch", batch)
        rt.process_blocks(2 * batch, 2)
        torch.cuda.synchronize(); t = time.time(); rt.process_blocks(1024, 2); dt = (time.time() - t) / 1024
        lv = rt.time_launches(2, 20)
        print("copies", copies, "batch", batch, "us/block %.2f" % (dt * 1e6), "launch us/block:", [round(1e3 * v / rt.last_time_batch, 2) for v in lv], "lds", rt.describe_plan()["max_lds_bytes"])
