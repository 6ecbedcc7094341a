"""One-off soak (GPU box): random graphs with wide mixers at random block sizes / launch-set sizes / grid rows / mixer
splits through the interpreter kernels (or, with a second argument 2, the specialised ones: every graph then costs its
kernel compilations), vs the reference engine. Usage: python tools/soak_geometry.py [seeds=120] [specialize=0]"""
import os as _os, sys as _sys; _R = _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))); _sys.path[:0] = [_R, _os.path.join(_R, 'tests')]
import sys
import numpy as np
import torch
import oracle
from elementary_amd import el
from elementary_amd.runtime import Runtime
from helpers import lcg_noise
from test_gpu_fuzz import random_graph

seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 120
spec = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rng = np.random.RandomState(7)
worst_all, bad = 0.0, []
for seed in range(seeds):
    bs = int(rng.choice([64, 128, 192, 256, 320, 448, 512] if spec else [64, 100, 128, 192, 256, 300, 320, 448, 512]))
    batch = int(rng.choice([3, 16, 64, 200]))
    rows = int(rng.choice([1, 5, 16, 64]))
    split = int(rng.choice([1, 2, 4, 8]))
    parts = random_graph(seed, n_nodes=30 + 10 * (seed % 5), n_roots=4)
    wide = [el.mul(0.1 + 0.01 * k, parts[k % len(parts)]) for k in range(8 + seed % 9)]          # >= 8 children from other islands
    roots = [el.add(*wide), el.mul(0.5, el.add(*[el.mul(0.2, p) for p in parts] * 3)), parts[0]]
    a = Runtime(48000.0, bs, device=0)
    for k, v in (("specialize", spec), ("batch_blocks", batch), ("stateless_rows", rows), ("mixer_split", split)):
        a.set_option(k, v)
    c = oracle.RefRuntime(48000.0, bs)
    assert a.render(*roots)["result"] == 0 and c.render(*roots)["result"] == 0
    nb = 26
    x = np.stack([np.stack([lcg_noise(bs, 11 + 7 * k + ch, 0.5) for ch in range(2)]) for k in range(nb)])
    xin = torch.from_numpy(x).cuda()
    out = torch.zeros((nb, 3, bs), dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    a.process_blocks(nb, 3, out_ptr=out.data_ptr(), in_ptr=xin.data_ptr(), num_inputs=2)
    got = out.cpu().numpy()
    ref = np.stack([c.process(x[k], 3, bs) for k in range(nb)])
    scale = max(1.0, float(np.abs(ref).max()))
    err = float(np.abs(got - ref).max()) / scale
    worst_all = max(worst_all, err)
    if spec and seed % 8 == 0:
        print(f"seed {seed}: bs {bs} batch {batch} shapes {a.stats()['spec_shapes']} launches {a.stats()['spec_launches']} err {err:.2e}", flush=True)
    if not (err <= 1e-6) or not np.isfinite(got).all():
        bad.append((seed, bs, batch, rows, split, err))
        print("MISMATCH", bad[-1], flush=True)
print(f"{seeds} graphs, worst relative error {worst_all:.3e}, mismatches: {bad}")
sys.exit(1 if bad else 0)
