#!/bin/bash
# Round validation on the GPU box: the whole -m gpu suite, then the per-call latency and C1 numbers.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; R=$PWD; O=$R/gpurun_out/${1:-r4m}; mkdir -p $O
timeout 1100 python -m pytest tests -m gpu -x -q < /dev/null > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -n 6 $O/pytest.log
timeout 120 python tools/process_latency.py 256 300 < /dev/null > $O/process_latency_c2.jsonl 2> $O/process_latency.err; cut -c1-260 $O/process_latency_c2.jsonl
timeout 120 python benchmarks/bench_configs.py c1 < /dev/null > $O/bench_c1.json 2> $O/bench_c1.err; cut -c1-400 $O/bench_c1.json
timeout 200 python benchmarks/bench_configs.py c5 < /dev/null > $O/bench_c5.json 2> $O/bench_c5.err; cut -c1-1500 $O/bench_c5.json
