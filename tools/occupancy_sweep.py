"""More islands than CUs: lane-packing (K voices per island) against TWO workgroups per CU (islands of <= 80 KB of LDS, kernels capped
at 128 VGPRs by `spec_waves_per_eu` = 4). Device-resident 128-block launch sets of the C2 synth at 256 / 512 / 1024 voices; the first
configuration of each size is the default, every other one must render the same samples bit for bit.
Usage: python tools/occupancy_sweep.py [sets]"""
import os as _os, sys as _sys; _R = _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))); _sys.path[:0] = [_R, _os.path.join(_R, 'tests')]
import sys
import time

import torch

from elementary_amd import graphs
from elementary_amd.runtime import Runtime

SETS = int(sys.argv[1]) if len(sys.argv) > 1 else 12
B = 128
CASES = [
    (256, "default", {}),
    (256, "K1 copies3 wpe4", {"pack_islands": 1, "pipeline_copies": 3, "spec_waves_per_eu": 4}),
    (512, "default (K2 fused)", {}),
    (512, "K1 copies3 wpe4", {"pack_islands": 1, "pipeline_copies": 3, "spec_waves_per_eu": 4}),
    (512, "K1 copies2 wpe4", {"pack_islands": 1, "pipeline_copies": 2, "spec_waves_per_eu": 4}),
    (512, "K1 copies3 (control: 141 VGPRs)", {"pack_islands": 1, "pipeline_copies": 3}),
    (1024, "default (K2 fused)", {}),
    (1024, "K2 copies2 wpe4", {"pack_islands": 2, "pipeline_copies": 2, "spec_waves_per_eu": 4}),
    (1024, "K1 copies3 wpe4", {"pack_islands": 1, "pipeline_copies": 3, "spec_waves_per_eu": 4}),
    (1024, "K1 copies2 wpe4", {"pack_islands": 1, "pipeline_copies": 2, "spec_waves_per_eu": 4}),
]
ref = {}
for voices, name, opts in CASES:
    rt = Runtime(graphs.C2_SAMPLE_RATE, 512, device=0)
    rt.set_option("specialize", 2)
    rt.set_option("batch_blocks", B)
    for k, v in opts.items():
        rt.set_option(k, v)
    assert rt.render(*graphs.c2_graph(voices=voices))["result"] == 0
    out = torch.empty((B, 2, 512), dtype=torch.float32, device="cuda")
    for _ in range(3):
        rt.process_blocks(B, 2, out_ptr=out.data_ptr())
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(SETS):
        rt.process_blocks(B, 2, out_ptr=out.data_ptr())
    torch.cuda.synchronize()
    us = 1e6 * (time.perf_counter() - t0) / (SETS * B)
    y = out.cpu().numpy()
    st, d = rt.stats(), rt.describe_plan()
    same = "(reference)" if voices not in ref else ("bit-identical" if (y == ref[voices]).all() else f"DIFFERS max {abs(y - ref[voices]).max():.3e}")
    ref.setdefault(voices, y)
    print(f"c2 {voices:5d} voices  {name:34s} {us:8.3f} us/block  {512 / us:7.1f} M samples/s  islands {st['num_islands']:5d}  lds {st['max_lds_bytes']:7d}  "
          f"K {d.get('pack_k')}  spec launches {st['spec_launches']}  {same}", flush=True)
