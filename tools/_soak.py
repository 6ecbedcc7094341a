"""Manual soak (not collected by pytest): long C2 render through the pipelined path vs the reference engine,
and a few hundred extra random graphs."""
import os as _os, sys as _sys; _R = _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))); _sys.path[:0] = [_R, _os.path.join(_R, 'tests')]

import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import oracle
from elementary_amd import graphs
from elementary_amd.runtime import Runtime
import test_gpu_fuzz as F
from helpers import lcg_noise

voices, blocks = 64, 3000
rt = Runtime(48000.0, 512); ref = oracle.RefRuntime(48000.0, 512)
roots = graphs.c2_graph(voices=voices)
assert rt.render(*roots)["result"] == 0 and ref.render(*roots)["result"] == 0
out = torch.zeros((blocks, 2, 512), dtype=torch.float32, device="cuda")
rt.process_blocks(blocks, 2, out_ptr=out.data_ptr())
got = out.cpu().numpy()
worst = 0.0
for k in range(blocks):
    worst = max(worst, float(np.abs(got[k] - ref.process(None, 2, 512)).max()))
print(f"C2 {voices} voices, {blocks} blocks pipelined vs reference: max abs err {worst:.3e}, batch launches {rt.stats()['batch_launches']}")
bad = 0
for seed in range(100, 300):
    nb, n_out = 14, min(3, 1 + seed % 5)
    x = np.stack([np.stack([lcg_noise(512, 11 + 7 * k + c, 0.5) for c in range(2)]) for k in range(nb)])
    a, b = Runtime(48000.0, 512), Runtime(48000.0, 512)
    b.set_option("batch_blocks", 3 + seed % 9)
    g = F.random_graph(seed, n_nodes=20 + 17 * (seed % 5), n_roots=1 + seed % 5)[:n_out]
    for r_ in (a, b):
        assert r_.render(*g)["result"] == 0
    got1 = np.stack([a.process(x[k], n_out, 512) for k in range(nb)])
    xin = torch.from_numpy(x).cuda(); o = torch.zeros((nb, n_out, 512), dtype=torch.float32, device="cuda"); torch.cuda.synchronize()
    b.process_blocks(nb, n_out, out_ptr=o.data_ptr(), in_ptr=xin.data_ptr(), num_inputs=2)
    if not np.array_equal(o.cpu().numpy(), got1):
        bad += 1; print("MISMATCH seed", seed, float(np.abs(o.cpu().numpy() - got1).max()))
print("random graphs 100..299: mismatches =", bad)
