"""Soak of feedback taps inside launch sets through the run-time specialised kernels (plan.cpp "taps inside launch sets",
island_ops.inc run_tapin / run_tapout): every in-set graph of tests/test_gpu_taps.py (+ the 8-loop bench graph) rendered for
thousands of blocks in 64-block sets and compared with the reference engine block by block. For every bad block the
report says where it sat in its launch set, which frames differ, and which neighbouring block's samples they carry — the
r03 mis-render (one block in ~150 on `cross` / `not_a_loop`) showed up as one wave's 128-frame range holding another
block's contents. Usage: python tools/tap_soak.py [blocks] [graph ...]   (one JSON line per graph)"""
import os as _os, sys as _sys; _R = _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))); _sys.path[:0] = [_R, _os.path.join(_R, 'tests')]
import json
import sys
import time

import numpy as np
import torch

from elementary_amd import el
from elementary_amd.runtime import Runtime
import oracle
from helpers import lcg_noise_fast as lcg_noise
import test_gpu_taps as T


def _bench_graph():
    def loop(k, x):
        fb = el.tapIn({"name": f"rv{k}"})
        body = el.lowpass(900.0 + 170.0 * k, 0.9, el.add(x, el.mul(0.7, el.sdelay({"size": 200 + 13 * k}, fb))))
        return el.tanh(el.tapOut({"name": f"rv{k}"}, body))
    x = el.in_({"channel": 0})
    return [el.add(*[loop(k, x) for k in range(0, 8, 2)]), el.add(*[loop(k, x) for k in range(1, 8, 2)])]


GRAPHS = {k: (v[0], v[1]) for k, v in T.CASES.items() if v[2]}
GRAPHS["eight_loops"] = (_bench_graph, 1)


def soak(name, nb, batch=64, spec=2, pattern="mixed"):
    roots_fn, n_in = GRAPHS[name]
    a = Runtime(44100.0, 512, device=0)
    a.set_option("batch_blocks", batch); a.set_option("specialize", spec)
    c = oracle.RefRuntime(44100.0, 512)
    roots = roots_fn()
    assert a.render(*roots)["result"] == 0 and c.render(*roots)["result"] == 0
    n_out = len(roots)
    x = np.stack([lcg_noise(nb * 512, 3 + ch, 0.5) for ch in range(n_in)])
    xin = torch.from_numpy(np.ascontiguousarray(x.reshape(n_in, nb, 512).transpose(1, 0, 2))).cuda()
    out = torch.empty((nb, n_out, 512), dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    # odd call sizes: sets of `batch`, ragged tails, single blocks in between
    cuts, k = [], 0
    sizes = [batch * 3 + 7, 1, batch, 2, batch * 5 + 1, 1, 1] if pattern == "mixed" else [batch * 4]
    i = 0
    while k < nb:
        n = min(sizes[i % len(sizes)] if i < 14 else batch * 16, nb - k)
        cuts.append((k, n)); k += n; i += 1
    t0 = time.perf_counter()
    for k0, n in cuts:
        a.process_blocks(n, n_out, out_ptr=out[k0].data_ptr(), in_ptr=xin[k0].data_ptr(), num_inputs=n_in)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    got = out.cpu().numpy()
    ref = np.stack([c.process(x[:, b * 512:(b + 1) * 512], n_out, 512) for b in range(nb)])
    scale = max(1.0, float(np.abs(ref).max()))
    err = np.abs(got - ref).max(axis=(1, 2))
    bad = np.nonzero(err > 1e-6 * scale)[0]
    st, plan = a.stats(), a.describe_plan()
    row = {"graph": name, "blocks": nb, "batch_blocks": batch, "specialize": spec, "spec_launches": st["spec_launches"], "batch_launches": st["batch_launches"],
           "taps_in_sets": plan["taps_in_sets"], "copies": [i["copies"] for i in plan["islands"]], "max_err": float(err.max()), "scale": scale,
           "bad_blocks": int(bad.size), "us_per_block": 1e6 * dt / nb}
    detail = []
    starts = np.array([k0 for k0, _ in cuts])
    for b in bad[:12]:
        call = int(np.searchsorted(starts, b, side="right") - 1)
        in_call = int(b - starts[call])
        fr = np.nonzero(np.abs(got[b] - ref[b]).max(axis=0) > 1e-6 * scale)[0]
        d = {"block": int(b), "call": call, "in_call": in_call, "in_set": in_call % batch, "frames": [int(fr.min()), int(fr.max()), int(fr.size)],
             "odd_even": [int((fr % 2 == 0).sum()), int((fr % 2 == 1).sum())], "err": float(err[b]), "next_block_err": float(err[b + 1]) if b + 1 < nb else None}
        # do the bad frames carry another block's output?
        for off in (-64, -3, -2, -1, 1, 2, 3):
            o = b + off
            if 0 <= o < nb and np.abs(got[b][:, fr] - ref[o][:, fr]).max() <= 1e-6 * scale:
                d["equals_ref_block"] = int(off)
        detail.append(d)
    row["bad_detail"] = detail
    return row


if __name__ == "__main__":
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("blocks", nargs="?", type=int, default=5000)
    ap.add_argument("graphs", nargs="*")
    ap.add_argument("--spec", type=int, default=2)
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--pattern", choices=["mixed", "sets"], default="mixed", help="sets: calls of 4 x batch blocks only (no single-block calls)")
    a_ = ap.parse_args()
    nb, names = a_.blocks, a_.graphs or sorted(GRAPHS)
    rc = 0
    for name in names:
        row = soak(name, nb, a_.batch, a_.spec, a_.pattern)
        print(json.dumps(row), flush=True)
        rc |= row["bad_blocks"] != 0 or (a_.spec != 0 and row["spec_launches"] == 0)
    sys.exit(rc)
