#!/bin/bash
# resident kernel (coalesced output stores), taps under host blocks > 512 frames, where the C1 call's time goes (kernel trace)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; R=$PWD; export TMPDIR=/tmp
O=$R/gpurun_out/${1:-r05h}; mkdir -p $O
(ELEMHIP_RESIDENT_TRACE=1 timeout 420 python -m pytest tests/test_gpu_resident.py -m gpu -q -rf -s --timeout 120 -p no:cacheprovider > $O/pytest_resident.log 2>&1; echo "rc=$?" >> $O/pytest_resident.log)
tail -12 $O/pytest_resident.log | cut -c1-250
(timeout 600 python -m pytest tests/test_gpu_taps.py tests/test_gpu_host_path.py -m gpu -q -rf --timeout 300 -p no:cacheprovider > $O/pytest_taps.log 2>&1; echo "rc=$?" >> $O/pytest_taps.log)
tail -12 $O/pytest_taps.log | cut -c1-250
timeout 400 python benchmarks/driver_configs.py c1 > $O/c1.json 2> $O/c1.err
python - <<PY
import json
j=json.load(open("$O/c1.json"))
print("c1 launch path", j.get("us_per_call"), "| resident", {k: v for k, v in j.get("resident_opt_in", {}).items() if k not in ("note", "parity")}, j.get("resident_opt_in", {}).get("parity", {}).get("ok"))
PY
python - > $O/c2_native.json 2> $O/c2_native.err <<PY
import json, sys, os
sys.path.insert(0, "benchmarks")
import bench_configs as bc
from elementary_amd import graphs
out = {}
for name, env in (("spec2", {"ELEMHIP_SPECIALIZE": "2"}), ("resident_spec1", {"ELEMHIP_SPECIALIZE": "1", "ELEMHIP_RESIDENT": "1", "ELEMHIP_RESIDENT_TRACE": "1"}), ("interp", {"ELEMHIP_SPECIALIZE": "0"})):
    out[name] = bc._native_host(graphs.c2_graph(voices=256, channels=2), graphs.C2_SAMPLE_RATE, blocks=2000, env=env)
print(json.dumps(out, indent=1))
PY
python -c "
import json; j=json.load(open('$O/c2_native.json')); print({k:(v or {}).get('us_p50') for k,v in j.items()})"
# the C1 call under the kernel trace: what the kernels themselves take
python -m elementary_amd.tools dump c1 /tmp/c1_batch.json > /dev/null 2>&1
cd /tmp && for spec in 0 2; do ELEMHIP_SPECIALIZE=$spec rocprofv3 --kernel-trace --stats -d $O/c1_trace_spec$spec -o c1 -- $R/examples/bench_cli /tmp/c1_batch.json 2000 44100 > $O/c1_trace_spec$spec.log 2>&1; done
cd $R
for spec in 0 2; do f=$(find $O/c1_trace_spec$spec -name "*kernel_stats.csv" | head -1); echo "spec $spec: $f"; head -6 "$f" | cut -c1-200; cp "$f" $O/c1_kernel_stats_spec$spec.csv 2>/dev/null; grep -h "Average iteration" $O/c1_trace_spec$spec.log; done
rm -rf $O/c1_trace_spec0 $O/c1_trace_spec2
