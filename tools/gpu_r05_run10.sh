#!/bin/bash
# sync_poll (elemhip_process spins on the epilogue's word instead of synchronising): A/B on C1 / C2 native hosts, then the whole GPU suite under the new default
cd "${GRAFT_REPO_ROOT:-/root/repo}"; R=$PWD; export TMPDIR=/tmp
O=$R/gpurun_out/${1:-r05j}; mkdir -p $O
python - <<PY
import sys
sys.path.insert(0, "$R")
from elementary_amd import graphs
from elementary_amd.reconciler import Renderer, batch_to_json
for name, roots in (("c1", graphs.c1_graph()), ("c2", graphs.c2_graph(voices=256, channels=2))):
    sent = []
    Renderer(lambda b: sent.append(b) or 0).render(*roots)
    open(f"/tmp/{name}_batch.json", "w").write(batch_to_json(sent[0]))
PY
for rep in 1 2; do for g in c1 c2; do sr=44100; [ $g = c2 ] && sr=48000
for poll in 0 1; do
  ELEMHIP_SPECIALIZE=2 ELEMHIP_SYNC_POLL=$poll $R/examples/bench_cli /tmp/${g}_batch.json 4000 $sr 2> $O/${g}_poll${poll}_$rep.json > /dev/null
  echo "$g sync_poll=$poll run $rep: $(cat $O/${g}_poll${poll}_$rep.json | cut -c60-200)"
done; done; done | tee $O/sync_poll_ab.txt
(timeout 2400 python -m pytest tests -m gpu -q -x -rf --timeout 900 -p no:cacheprovider > $O/pytest_gpu.log 2>&1; echo "rc=$?" >> $O/pytest_gpu.log)
tail -5 $O/pytest_gpu.log | cut -c1-250
