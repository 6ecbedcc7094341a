#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; R=$PWD; export TMPDIR=/tmp
O=$R/gpurun_out/${1:-r05f}; mkdir -p $O
(timeout 600 python -m pytest tests/test_gpu_product_mode.py tests/test_gpu_churn.py tests/test_gpu_spec.py tests/test_gpu_multiproc.py -m gpu -q -rf --timeout 600 -p no:cacheprovider > $O/pytest_sel.log 2>&1; echo "rc=$?" >> $O/pytest_sel.log)
for k in 1 2 3; do timeout 300 python benchmarks/driver_configs.py c5 > $O/c5_run$k.json 2> $O/c5_run$k.err; done
tail -3 $O/pytest_sel.log; for k in 1 2 3; do python -c "import json; j=json.load(open('$O/c5_run$k.json')); print(j['counted']['commit_to_first_block_ms'], j['counted']['commit_call_ms'], j['counted']['first_block_ms'], j['parity']['ok'])"; done
