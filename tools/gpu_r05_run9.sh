#!/bin/bash
# where a synchronous elemhip_process call's time goes: the C1 and C2 native hosts under the kernel trace (durations of the kernels themselves)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; R=$PWD; export TMPDIR=/tmp
O=$R/gpurun_out/${1:-r05i}; mkdir -p $O
python -m elementary_amd.tools dump c1 /tmp/c1_batch.json > /dev/null 2>&1
python - <<PY
import sys
sys.path.insert(0, "$R")
from elementary_amd import graphs
from elementary_amd.reconciler import Renderer, batch_to_json
sent = []
Renderer(lambda b: sent.append(b) or 0).render(*graphs.c2_graph(voices=256, channels=2))
open("/tmp/c2_batch.json", "w").write(batch_to_json(sent[0]))
sent = []
Renderer(lambda b: sent.append(b) or 0).render(*graphs.c1_graph())
open("/tmp/c1_batch.json", "w").write(batch_to_json(sent[0]))
PY
for g in c1 c2; do sr=44100; [ $g = c2 ] && sr=48000
for spec in 0 2; do
  ELEMHIP_SPECIALIZE=$spec $R/examples/bench_cli /tmp/${g}_batch.json 2000 $sr 2> $O/${g}_plain_spec$spec.json > /dev/null
  (cd /tmp && ELEMHIP_SPECIALIZE=$spec rocprofv3 --kernel-trace --stats --output-format csv -d $O/tr_${g}_$spec -- $R/examples/bench_cli /tmp/${g}_batch.json 2000 $sr > $O/${g}_trace_spec$spec.log 2>&1)
  f=$(find $O/tr_${g}_$spec -name "*kernel_stats.csv" | head -1); cp "$f" $O/${g}_sync_kernel_stats_spec$spec.csv 2>/dev/null
  echo "== $g spec $spec: plain $(grep -o '"us_p50": [0-9.]*' $O/${g}_plain_spec$spec.json) | under the trace: $(grep -h 'Average iteration' $O/${g}_trace_spec$spec.log)"
  head -7 $O/${g}_sync_kernel_stats_spec$spec.csv | cut -c1-160
  rm -rf $O/tr_${g}_$spec
done; done
