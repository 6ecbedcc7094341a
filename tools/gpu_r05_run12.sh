#!/bin/bash
# the -m gpu suite two more ways: interpreter kernels pinned; stream synchronise instead of the epilogue's word
cd "${GRAFT_REPO_ROOT:-/root/repo}"; R=$PWD; export TMPDIR=/tmp
O=$R/gpurun_out/${1:-r05l}; mkdir -p $O
(ELEMHIP_TEST_SPECIALIZE=0 timeout 1500 python -m pytest tests -m gpu -q -rf --timeout 900 -p no:cacheprovider > $O/pytest_spec0.log 2>&1; echo "rc=$?" >> $O/pytest_spec0.log)
tail -4 $O/pytest_spec0.log | cut -c1-250
(ELEMHIP_SYNC_POLL=0 timeout 1500 python -m pytest tests -m gpu -q -rf --timeout 900 -p no:cacheprovider > $O/pytest_nopoll.log 2>&1; echo "rc=$?" >> $O/pytest_nopoll.log)
tail -4 $O/pytest_nopoll.log | cut -c1-250
