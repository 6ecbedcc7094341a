#!/bin/bash
# convolve A/B on the GPU box: the conv tests, then C3 with the LDS-tiled MAC and direct I/O on / off, then the C3 profile passes
cd "${GRAFT_REPO_ROOT:-/root/repo}"; R=$PWD; export TMPDIR=/tmp
O=$R/gpurun_out/${1:-r05d}; mkdir -p $O
(timeout 900 python -m pytest tests/test_gpu_convolve.py tests/test_gpu_host_path.py tests/test_gpu_golden.py -m gpu -q -rf --timeout 600 -p no:cacheprovider > $O/pytest_conv.log 2>&1; echo "rc=$?" >> $O/pytest_conv.log)
for v in "1 1" "0 1" "1 0" "0 0"; do set -- $v; ELEMHIP_C3_OPTS="conv_long_mac_lds=$1,conv_direct_io=$2" timeout 200 python benchmarks/driver_configs.py c3 --gpu-only > $O/c3_lds$1_dio$2.json 2> $O/c3_lds$1_dio$2.err; done
(timeout 300 python benchmarks/driver_configs.py c3 > $O/c3_full.json 2> $O/c3_full.err)
bash profiles/collect_r05.sh $(basename $O) c3 > $O/collect.log 2>&1
tail -4 $O/pytest_conv.log; for f in $O/c3_lds*.json; do echo $f; python -c "import json; j=json.load(open('$f')); print(j['us_per_block'], j['launch_us_per_step'], j.get('conv_long_sets'))"; done; python -c "import json; j=json.load(open('$O/c3_full.json')); print(j['us_per_block'], j['parity']['ok'], j['parity']['max_abs_err'])"; grep -E "^trace|own step" $O/collect.log | head -14
