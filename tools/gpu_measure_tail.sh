#!/bin/bash
# The single-block / live-graph measurements of the round on the GPU box (no test suite): per-call latency (Python and native
# host), C1, C5 and its commit breakdown, render-call latency by blocks since a commit.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; R=$PWD; O=$R/gpurun_out/${1:-r4q}; mkdir -p $O
for w in 256 c1 floor; do timeout 120 python tools/process_latency.py $w 300 < /dev/null > $O/process_latency_$w.jsonl 2> /dev/null; sed -n 3p $O/process_latency_$w.jsonl | cut -c1-200; done
timeout 200 python benchmarks/bench_configs.py c1 < /dev/null > $O/bench_c1.json 2> $O/bench_c1.err; cut -c1-300 $O/bench_c1.json
timeout 200 python benchmarks/bench_configs.py c5 < /dev/null > $O/bench_c5.json 2> $O/bench_c5.err; cut -c1-900 $O/bench_c5.json
timeout 200 python tools/c5_commit_breakdown.py 230 < /dev/null > $O/c5_commit_breakdown.json 2> /dev/null; cut -c1-600 $O/c5_commit_breakdown.json
timeout 200 python tools/c5_render_after_commit.py < /dev/null > $O/c5_render_after_commit.json 2> /dev/null; cut -c1-300 $O/c5_render_after_commit.json
# kernel trace of the native host's synchronous calls (C1 and the 256-voice graph): what one elemhip_process costs on the device
python - <<'PY' > /dev/null 2>&1
import sys, os
sys.path.insert(0, os.getcwd())
from elementary_amd import graphs
from elementary_amd.reconciler import Renderer, batch_to_json
for name, roots in (("c1", graphs.c1_graph()), ("c2", graphs.c2_graph())):
    sent = []
    Renderer(lambda b: sent.append(b) or 0).render(*roots)
    open(f"/tmp/{name}_batch.json", "w").write(batch_to_json(sent[0]))
PY
for g in c1 c2; do
  sr=44100; [ $g = c2 ] && sr=48000
  (cd /tmp && export TMPDIR=/tmp && ELEMHIP_SPECIALIZE=2 timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_cli_$g -- $R/examples/bench_cli /tmp/${g}_batch.json 3000 $sr < /dev/null > $O/prof_cli_$g.log 2>&1)
  f=$(ls $O/prof_cli_$g/*/*kernel_stats.csv 2>/dev/null | tail -n 1)
  [ -n "$f" ] && cp "$f" $O/cli_${g}_kernel_stats.csv && head -n 8 "$f" | cut -c1-160
  find $O/prof_cli_$g -name "*kernel_trace.csv" -size +4M -delete
done
