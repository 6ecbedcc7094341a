#!/bin/bash
# resident kernel: its tests first (bounded), then C1 with the resident leg, then C2's native-host call three ways
cd "${GRAFT_REPO_ROOT:-/root/repo}"; R=$PWD; export TMPDIR=/tmp
O=$R/gpurun_out/${1:-r05g}; mkdir -p $O
(ELEMHIP_RESIDENT_TRACE=1 timeout 420 python -m pytest tests/test_gpu_resident.py -m gpu -q -rf -s --timeout 120 -p no:cacheprovider > $O/pytest_resident.log 2>&1; echo "rc=$?" >> $O/pytest_resident.log)
tail -25 $O/pytest_resident.log
if grep -q "rc=0" $O/pytest_resident.log || [ "$2" = "force" ]; then
  timeout 400 python benchmarks/driver_configs.py c1 > $O/c1.json 2> $O/c1.err
  python - <<PY
import json
j=json.load(open("$O/c1.json"))
print("c1 launch path", j.get("us_per_call"), "| resident", {k: v for k, v in j.get("resident_opt_in", {}).items() if k != "note"})
PY
  python - > $O/c2_native.json 2> $O/c2_native.err <<PY
import json, sys, os
sys.path.insert(0, "benchmarks")
import bench_configs as bc
from elementary_amd import graphs
out = {}
for name, env in (("spec2", {"ELEMHIP_SPECIALIZE": "2"}), ("resident_spec1", {"ELEMHIP_SPECIALIZE": "1", "ELEMHIP_RESIDENT": "1"}), ("interp", {"ELEMHIP_SPECIALIZE": "0"}), ("resident_interp", {"ELEMHIP_SPECIALIZE": "0", "ELEMHIP_RESIDENT": "1"})):
    out[name] = bc._native_host(graphs.c2_graph(voices=256, channels=2), graphs.C2_SAMPLE_RATE, blocks=2000, env=env)
print(json.dumps(out, indent=1))
PY
  cat $O/c2_native.json | head -40
fi
