#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; R=$PWD; export TMPDIR=/tmp
O=$R/gpurun_out/${1:-r05o}; mkdir -p $O
(timeout 900 python -m pytest tests/test_gpu_host_path.py tests/test_gpu_taps.py tests/test_gpu_spec.py -m gpu -q -rf --timeout 600 -p no:cacheprovider > $O/pytest_sel.log 2>&1; echo "rc=$?" >> $O/pytest_sel.log)
tail -25 $O/pytest_sel.log | cut -c1-250
