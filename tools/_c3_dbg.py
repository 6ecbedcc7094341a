import os as _os, sys as _sys; _R = _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))); _sys.path[:0] = [_R, _os.path.join(_R, 'tests')]
import sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from elementary_amd import graphs, el
from elementary_amd.runtime import Runtime
ch, blocks = 2, 24
def mk(conv=True):
    rt = Runtime(48000.0, 512, device=0)
    for c in range(ch): rt.add_shared_resource(f"ir{c}", graphs.c3_impulse_response(c))
    roots = graphs.c3_graph(ch) if conv else [el.mul(2.0, el.in_({"channel": c})) for c in range(ch)]
    assert rt.render(*roots)["result"] == 0
    return rt
x = graphs.c3_input(ch, blocks * 512)
for conv in (False, True):
    rt = mk(conv)
    ref = np.stack([rt.process(x[:, k*512:(k+1)*512], ch, 512) for k in range(blocks)])
    rt2 = mk(conv)
    xin = torch.from_numpy(np.ascontiguousarray(x.reshape(ch, blocks, 512).transpose(1, 0, 2))).cuda()
    out = torch.empty((blocks, ch, 512), dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    rt2.process_blocks(blocks, ch, out_ptr=out.data_ptr(), in_ptr=xin.data_ptr(), num_inputs=ch)
    o = out.cpu().numpy()
    print("conv" if conv else "gain", [float(np.abs(o[k]-ref[k]).max()) for k in range(blocks)])
