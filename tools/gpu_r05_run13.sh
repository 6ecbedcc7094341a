#!/bin/bash
# soaks of the final tree: the C5 mutation stream (every block and gc result), taps inside launch sets, the full C2 graph
cd "${GRAFT_REPO_ROOT:-/root/repo}"; R=$PWD; export TMPDIR=/tmp
O=$R/gpurun_out/${1:-r05n}; mkdir -p $O
(timeout 900 python tools/soak_c5.py 300 > $O/soak_c5_300.txt 2>&1; echo "rc=$?" >> $O/soak_c5_300.txt); tail -3 $O/soak_c5_300.txt | cut -c1-300
(timeout 900 python tools/tap_soak.py 5000 > $O/tap_soak_5000.jsonl 2>&1; echo "rc=$?" >> $O/tap_soak_5000.jsonl); tail -8 $O/tap_soak_5000.jsonl | cut -c1-200
(timeout 900 python tools/soak_c2.py 5120 > $O/soak_c2_5120.txt 2>&1; echo "rc=$?" >> $O/soak_c2_5120.txt); tail -3 $O/soak_c2_5120.txt | cut -c1-300
