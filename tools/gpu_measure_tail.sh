#!/bin/bash
# The single-block / live-graph measurements of the round on the GPU box (no test suite): per-call latency (Python and native
# host), C1, C5 and its commit breakdown, render-call latency by blocks since a commit.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; R=$PWD; O=$R/gpurun_out/${1:-r4q}; mkdir -p $O
for w in 256 c1 floor; do timeout 120 python tools/process_latency.py $w 300 < /dev/null > $O/process_latency_$w.jsonl 2> /dev/null; sed -n 3p $O/process_latency_$w.jsonl | cut -c1-200; done
timeout 200 python benchmarks/bench_configs.py c1 < /dev/null > $O/bench_c1.json 2> $O/bench_c1.err; cut -c1-300 $O/bench_c1.json
timeout 200 python benchmarks/bench_configs.py c5 < /dev/null > $O/bench_c5.json 2> $O/bench_c5.err; cut -c1-900 $O/bench_c5.json
timeout 200 python tools/c5_commit_breakdown.py 230 < /dev/null > $O/c5_commit_breakdown.json 2> /dev/null; cut -c1-600 $O/c5_commit_breakdown.json
timeout 200 python tools/c5_render_after_commit.py < /dev/null > $O/c5_render_after_commit.json 2> /dev/null; cut -c1-300 $O/c5_render_after_commit.json
