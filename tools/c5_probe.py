"""dev tool: per-batch timing of the config-5 mutation stream on the GPU (apply, first block, stats deltas)."""
import os as _os, sys as _sys; _R = _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))); _sys.path[:0] = [_R, _os.path.join(_R, 'tests'), _os.path.join(_R, 'benchmarks')]
import sys, time
import torch
from bench_configs import _c5_batches
from elementary_amd.runtime import Runtime
spec = int(sys.argv[1]) if len(sys.argv) > 1 else 2
texts, creates, sizes = _c5_batches(128, 40)
rt = Runtime(48000.0, 512, device=0)
rt.set_option("specialize", spec)
out = torch.empty((64, 2, 512), dtype=torch.float32, device="cuda")
for k, t in enumerate(texts):
    t0 = time.time(); rc = rt.apply_instructions_json(t); t1 = time.time()
    rt.process_blocks(1, 2, out_ptr=out.data_ptr()); t2 = time.time()
    rt.process_blocks(64, 2, out_ptr=out.data_ptr()); t3 = time.time()
    st = rt.stats()
    print(k, rc, "apply %.1f ms first block %.1f ms 64 blocks %.1f ms | plan %.1f jit %.1f shapes %d islands %d hbm %d captures %d" % (
        1e3*(t1-t0), 1e3*(t2-t1), 1e3*(t3-t2), st["last_plan_build_ms"], st["last_jit_wait_ms"], st["spec_shapes"], st["spec_islands"], st["num_hbm_buffers"], st["graph_captures"]), flush=True)
    if k % 16 == 15:
        t0 = time.time(); n = len(rt.gc()); print("  gc", n, "%.1f ms" % (1e3*(time.time()-t0)))
