#!/bin/bash
# Round measurements on the GPU box: the driver-shaped bench line, the C4 line with its CPU baseline and parity, C3, and the
# rocprofv3 passes (kernel trace + FETCH_SIZE + WRITE_SIZE, separate runs) of all three.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; R=$PWD; O=$R/gpurun_out/${1:-r4p}; mkdir -p $O
timeout 200 python -m pytest tests/test_gpu_product_mode.py -q -x < /dev/null 2>&1 | tail -n 3 | tee $O/pytest_product_mode.txt
timeout 300 python bench.py --steps 20 --warmup 5 < /dev/null > $O/bench_c2_n1_driver_shaped.json 2> $O/bench_c2.err; cut -c1-300 $O/bench_c2_n1_driver_shaped.json
timeout 400 python bench.py --workload c4 < /dev/null > $O/bench_c4_n1_workload.json 2> $O/bench_c4.err; cut -c1-200 $O/bench_c4_n1_workload.json; tail -n 2 $O/bench_c4.err
timeout 200 python benchmarks/bench_configs.py c3 < /dev/null > $O/bench_c3.json 2> $O/bench_c3.err; cut -c1-200 $O/bench_c3.json
timeout 120 python tools/tap_loop_bench.py < /dev/null > $O/tap_loop_bench.jsonl 2>/dev/null; tail -n 2 $O/tap_loop_bench.jsonl | cut -c1-200
timeout 1000 bash profiles/collect_r04.sh ${1:-r4p} c2 c3 c4 < /dev/null 2>&1 | tail -n 45
