#!/bin/bash
# Run ON THE GPU BOX (gpurun): second pass of the round — the -m gpu suite as the driver runs it (product default), the default bench
# line, and the rocprofv3 passes of C3 (new long-partition kernels), C2, C4 and the one-shape C4 (profile reconciliation).
cd "${GRAFT_REPO_ROOT:-/root/repo}"; R=$PWD; export TMPDIR=/tmp
O=$R/gpurun_out/${1:-r05b}; mkdir -p $O
(timeout 1200 python -m pytest tests -m gpu -q -rf --timeout 600 -p no:cacheprovider > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log)
(timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "rc=$?" >> $O/bench.err)
bash profiles/collect_r05.sh ${1:-r05b} ${2:-c3 c2 c4 c4x64} > $O/collect.log 2>&1
tail -6 $O/pytest.log; head -c 300 $O/bench.json; echo; grep -E "^trace|own step" $O/collect.log | head -60
