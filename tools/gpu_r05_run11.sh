#!/bin/bash
# final pass: the -m gpu suite as the driver runs it, smoke, the default bench line (with its configs)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; R=$PWD; export TMPDIR=/tmp
O=$R/gpurun_out/${1:-r05k}; mkdir -p $O
(timeout 1500 python -m pytest tests -m gpu -x -q -rf --timeout 900 -p no:cacheprovider > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log)
(timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; echo "rc=$?" >> $O/smoke.log)
(timeout 1200 python bench.py > $O/bench.json 2> $O/bench.err; echo "rc=$?" >> $O/bench.err)
tail -4 $O/pytest.log; tail -2 $O/smoke.log; head -c 400 $O/bench.json; echo; tail -2 $O/bench.err
