"""A/B of option `fuse_epilogue` for the synchronous call (examples/bench_cli, the reference's cli/Benchmark.cpp protocol, native host):
C1 and the 256-voice C2 graph, ELEMHIP_FUSE_EPILOGUE = 0 / 1, back to back, twice. One JSON line per run."""
import os as _os, sys as _sys; _R = _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))); _sys.path[:0] = [_R, _os.path.join(_R, 'benchmarks')]
import json
import os
import tempfile

import torch  # noqa: F401

import driver_configs as D
from elementary_amd import graphs

with tempfile.TemporaryDirectory() as d:
    for rep in range(2):
        for name, roots, sr in (("c1", graphs.c1_graph(), graphs.C1_SAMPLE_RATE), ("c2", graphs.c2_graph(voices=256), graphs.C2_SAMPLE_RATE)):
            for fuse in (0, 1):
                r = D._native_run(roots, sr, 4000, 2, os.path.join(d, "all.f32"), env_extra={"ELEMHIP_FUSE_EPILOGUE": str(fuse)})
                print(json.dumps({"graph": name, "fuse_epilogue": fuse, **{k: r.get(k) for k in ("us_mean", "us_p50", "us_p99", "error")}}), flush=True)
