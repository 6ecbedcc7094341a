import sys
sys.path.insert(0, '.')
from elementary_amd import graphs
from elementary_amd.runtime import Runtime
for spec in (0, 2):
    rt = Runtime(44100.0, 512, device=0)
    rt.set_option("specialize", spec)
    assert rt.render(*graphs.c1_graph())["result"] == 0
    for _ in range(20): rt.process(None, 2, 512)
    ms = rt.time_launches(2, 200)
    st = rt.stats()
    print("C1 spec", spec, "levels", st["num_levels"], "islands", st["num_islands"], "lds", st["max_lds_bytes"], "launch us", [round(1e3 * x, 2) for x in ms])
