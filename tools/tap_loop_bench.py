"""A feedback loop through a tap (lowpass + 300-frame delay inside the loop, 8 such loops under two roots) rendered by
elemhip_process_blocks: block-at-a-time (batch_blocks = 1: what every plan with a tapOut got before taps could be rendered
inside launch sets) vs 64-block launch sets through the interpreter kernel vs 64- / 256-block sets through the run-time specialised
kernels (r04: tap islands have them; the hand-over stays in the tapOut's LDS slot). Usage: python tools/tap_loop_bench.py [blocks]"""
import os as _os, sys as _sys; _R = _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))); _sys.path[:0] = [_R, _os.path.join(_R, 'tests')]
import json
import sys
import time

import numpy as np
import torch

from elementary_amd import el
from elementary_amd.runtime import Runtime

blocks = int(sys.argv[1]) if len(sys.argv) > 1 else 2048


def loop(k, x):
    fb = el.tapIn({"name": f"rv{k}"})
    body = el.lowpass(900.0 + 170.0 * k, 0.9, el.add(x, el.mul(0.7, el.sdelay({"size": 200 + 13 * k}, fb))))
    return el.tanh(el.tapOut({"name": f"rv{k}"}, body))


def graph():
    x = el.in_({"channel": 0})
    return [el.add(*[loop(k, x) for k in range(0, 8, 2)]), el.add(*[loop(k, x) for k in range(1, 8, 2)])]


rows = []
for batch, spec in ((1, 0), (64, 0), (64, 2), (256, 2)):
    rt = Runtime(48000.0, 512, device=0)
    rt.set_option("batch_blocks", batch)
    rt.set_option("specialize", spec)
    assert rt.render(*graph())["result"] == 0
    x = torch.rand((256, 1, 512), device="cuda") - 0.5
    out = torch.empty((256, 2, 512), dtype=torch.float32, device="cuda")
    rt.process_blocks(256, 2, out_ptr=out.data_ptr(), in_ptr=x.data_ptr(), num_inputs=1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(blocks // 256):
        rt.process_blocks(256, 2, out_ptr=out.data_ptr(), in_ptr=x.data_ptr(), num_inputs=1)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / (blocks // 256 * 256)
    plan = rt.describe_plan()
    rows.append({"batch_blocks": batch, "specialize": spec, "spec_launches": rt.stats()["spec_launches"], "us_per_block": 1e6 * dt, "samples_per_s": 512 / dt, "batch_launches": rt.stats()["batch_launches"],
                 "taps_in_sets": plan["taps_in_sets"], "islands": plan["num_islands"], "levels": plan["num_levels"]})
    print(json.dumps(rows[-1]), flush=True)
print(json.dumps({"speedup_vs_block_at_a_time": [rows[0]["us_per_block"] / r["us_per_block"] for r in rows]}))
