#!/bin/bash
# Run ON THE GPU BOX (gpurun): the round's first validation pass — the -m gpu suite under the pinned interpreter default and under
# the product default, the default bench line (with the configs sub-records), the C4 biquad-form A/B.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; R=$PWD; export TMPDIR=/tmp
O=$R/gpurun_out/${1:-r05a}; mkdir -p $O
rocminfo 2>/dev/null | grep -m3 -E "Marketing Name|Compute Unit" > $O/box.txt; nproc >> $O/box.txt; lscpu | grep -m1 "Model name" >> $O/box.txt
(timeout 1200 python -m pytest tests -m gpu -q -rf --timeout 600 -p no:cacheprovider > $O/pytest_pin0.log 2>&1; echo "rc=$?" >> $O/pytest_pin0.log)
(timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "rc=$?" >> $O/bench.err)
for f in 0 1 2; do (timeout 200 python bench.py --workload c4 --steps 12 --warmup 3 --no-cpu-baseline --opt biquad_form=$f > $O/c4_form$f.json 2> $O/c4_form$f.err); done
(ELEMHIP_TEST_SPECIALIZE=1 timeout 1200 python -m pytest tests -m gpu -q -rf --timeout 600 -p no:cacheprovider > $O/pytest_product.log 2>&1; echo "rc=$?" >> $O/pytest_product.log)
tail -5 $O/pytest_pin0.log; tail -5 $O/pytest_product.log; head -c 600 $O/bench.json; echo; for f in 0 1 2; do python -c "import json,sys; j=json.load(open('$O/c4_form$f.json')); print('c4 form $f', j['value'], j['ms_per_step'])" 2>&1 | tail -1; done
