"""One-off soak (GPU box): the config-5 mutation stream (128 live voices, one replaced per batch, gc every 16 batches)
through the product defaults — background kernel specialisation, 64-block launch sets — with 1 to 40 blocks between
batches, every block and every gc() result vs the reference engine. Usage: python tools/soak_c5.py [batches=300]"""
import os as _os, sys as _sys; _R = _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))); _sys.path[:0] = [_R, _os.path.join(_R, 'tests')]
import sys, time
import numpy as np
import torch
import oracle
from elementary_amd import el, graphs
from elementary_amd.runtime import Runtime

n_batches = int(sys.argv[1]) if len(sys.argv) > 1 else 300
heap_dwords = int(sys.argv[2]) if len(sys.argv) > 2 else 0        # > 0: a program heap that keeps running out (plan.cpp ProgHeap)


def voices_graph(ids):
    outs = []
    for c in range(2):
        vs = [graphs.c2_voice(v) for k, v in enumerate(ids) if k % 2 == c]
        outs.append(el.add(*vs))
    return outs


a = Runtime(graphs.C2_SAMPLE_RATE, 512, device=0)
a.set_option("specialize", 1)
if heap_dwords:
    a.set_option("prog_heap_dwords", heap_dwords)
c = oracle.RefRuntime(graphs.C2_SAMPLE_RATE, 512)
rng = np.random.RandomState(3)
ids, nxt, worst, blocks, pruned_total = list(range(128)), 128, 0.0, 0, 0
out = torch.zeros((64, 2, 512), dtype=torch.float32, device="cuda")
t0 = time.time()
for batch in range(n_batches):
    if batch:
        ids[(batch * 37) % 128] = nxt
        nxt += 1
    roots = voices_graph(ids)
    assert a.render(*roots)["result"] == 0 and c.render(*roots)["result"] == 0
    nb = int(rng.choice([1, 2, 3, 5, 8, 13, 40]))
    ref = np.stack([c.process(None, 2, 512) for _ in range(nb)])
    if batch % 3 == 0:
        got = np.stack([a.process(None, 2, 512) for _ in range(nb)])
    else:
        torch.cuda.synchronize()
        a.process_blocks(nb, 2, out_ptr=out.data_ptr())
        got = out[:nb].cpu().numpy()
    err = float(np.abs(got - ref).max())
    worst = max(worst, err)
    blocks += nb
    if err > 1e-6:
        print("MISMATCH at batch", batch, err); sys.exit(1)
    if batch % 16 == 15:
        pa, pc = sorted(a.gc()), sorted(c.gc())
        assert pa == pc, (batch, len(pa), len(pc))
        pruned_total += len(pa)
    if batch % 50 == 49:
        st = a.stats()
        print(f"batch {batch + 1}: {blocks} blocks, worst {worst:.2e}, spec launches {st['spec_launches']}, plans {st['plans_built']}, pruned {pruned_total}, {time.time() - t0:.0f} s", flush=True)
info = a.describe_plan()
print("done: worst abs err", worst, {k: info[k] for k in info if k.startswith("plan_")})
