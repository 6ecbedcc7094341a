import sys, time; sys.path.insert(0, '.')
import numpy as np, torch
from elementary_amd import graphs
from elementary_amd.runtime import Runtime
rt = Runtime(48000.0, 512, device=0); rt.set_option("specialize", 2); rt.set_option("batch_blocks", 256)
assert rt.render(*graphs.c2_graph())["result"] == 0
out = np.zeros((2, 16 * 256 * 512), np.float32)
dev = torch.zeros((256, 2, 512), dtype=torch.float32, device="cuda")
for prof in (0, 1):
    rt.set_option("profile_launches", prof)
    for sets in (1, 2, 4, 8, 16):
        rt.process_blocks_host(None, 2, sets * 256 * 512, out=out)
        t0 = time.perf_counter(); rt.process_blocks_host(None, 2, sets * 256 * 512, out=out); t1 = time.perf_counter()
        t2 = time.perf_counter()
        for _ in range(sets): rt.process_blocks(256, 2, out_ptr=dev.data_ptr())
        t3 = time.perf_counter()
        print(f"profile {prof} sets {sets:2d}: host {1e3*(t1-t0):7.3f} ms ({1e6*(t1-t0)/(sets*256):6.3f} us/block)  device {1e3*(t3-t2):7.3f} ms ({1e6*(t3-t2)/(sets*256):6.3f} us/block)", flush=True)
