import os as _os, sys as _sys; _R = _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))); _sys.path[:0] = [_R, _os.path.join(_R, 'tests')]
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import conv_cases as C
from elementary_amd import graphs
from elementary_amd.runtime import Runtime

for name in sorted(C.SCENARIOS):
    y = C.run_scenario(lambda sr, bs: Runtime(sr, bs, device=0), name).astype(np.float64)
    print(name, "vs wasm %.3g  vs exact %.3g" % (np.abs(y - C.golden(name)).max(), np.abs(y - C.exact_model(name)).max()))

ch = 8
rt = Runtime(48000.0, 512, device=0)
for c in range(ch):
    rt.add_shared_resource(f"ir{c}", graphs.c3_impulse_response(c))
assert rt.render(*graphs.c3_graph(ch))["result"] == 0
print(rt.describe_plan()["level_sizes"], rt.describe_plan()["conv_workgroups"])
blocks = 2048
x = torch.from_numpy(np.ascontiguousarray(graphs.c3_input(ch, 64 * 512).reshape(ch, 64, 512).transpose(1, 0, 2))).cuda().repeat(blocks // 64, 1, 1).contiguous()
out = torch.empty((blocks, ch, 512), dtype=torch.float32, device="cuda")
torch.cuda.synchronize()
for rep in range(3):
    t = time.time()
    rt.process_blocks(blocks, ch, out_ptr=out.data_ptr(), in_ptr=x.data_ptr(), num_inputs=ch)
    dt = time.time() - t
    print("process_blocks: %.2f us/block, %.2f Msamples/s per channel-set" % (dt / blocks * 1e6, blocks * 512 / dt / 1e6))
print("levels ms:", rt.time_launches(ch, 200))
