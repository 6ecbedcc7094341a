"""One-off soak: the full C2 graph through 256-block launch sets of the specialised kernels vs the reference engine.
Usage (GPU box): python tools/soak_c2.py [blocks=5120]"""
import os as _os, sys as _sys; _R = _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))); _sys.path[:0] = [_R, _os.path.join(_R, 'tests')]
import sys, time
import numpy as np
import torch
import oracle
from elementary_amd import graphs
from elementary_amd.runtime import Runtime

nb = int(sys.argv[1]) if len(sys.argv) > 1 else 5120
a = Runtime(graphs.C2_SAMPLE_RATE, 512, device=0)
a.set_option("specialize", 2); a.set_option("batch_blocks", 256)
c = oracle.RefRuntime(graphs.C2_SAMPLE_RATE, 512)
roots = graphs.c2_graph()
assert a.render(*roots)["result"] == 0 and c.render(*roots)["result"] == 0
worst, done = 0.0, 0
out = torch.zeros((1024, 2, 512), dtype=torch.float32, device="cuda")
t0 = time.time()
while done < nb:
    n = min(1024, nb - done)
    torch.cuda.synchronize()
    a.process_blocks(n, 2, out_ptr=out.data_ptr())
    got = out[:n].cpu().numpy()
    ref = np.stack([c.process(None, 2, 512) for _ in range(n)])
    worst = max(worst, float(np.abs(got - ref).max()))
    done += n
    print(f"{done} blocks, max abs err so far {worst:.3e}, {time.time() - t0:.0f} s", flush=True)
st = a.stats()
print("spec launches", st["spec_launches"], "batch launches", st["batch_launches"])
assert worst <= 1e-6
