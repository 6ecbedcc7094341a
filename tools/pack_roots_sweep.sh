#!/bin/bash
# Run ON THE GPU BOX: C4 render jobs (a root each) at 256 / 512 / 1024 jobs per GPU, one job per workgroup (rounds of 256) vs
# lane-packed across roots (`pack_roots` = 1, K = ceil(jobs / CUs) up to 4). Each line carries the bench's own parity check.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; out=$PWD/gpurun_out/${1:-r4q}; mkdir -p $out
run() { tag=$1; shift; timeout 200 python bench.py --workload c4 --no-cpu-baseline --steps 6 --warmup 2 --batch-blocks 128 --device-resident "$@" < /dev/null > $out/$tag.json 2> $out/$tag.err
  timeout 20 python - $out/$tag.json $tag <<'PY' | tee -a $out/summary.txt
import json,sys
try:
    b=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); c=b["config"]
    print("%-18s %8.3f G instance-samples/s  us/block-step %8.3f  per-job ns/block %7.1f  islands %s  parity %s" % (sys.argv[2], b["value"]/1e9, b["us_per_block_step"], 1e3*b["us_per_block_step"]/c["instances_per_gpu"], c.get("islands"), (b.get("parity") or {}).get("ok")))
except Exception as e: print(sys.argv[2], "failed", e)
PY
}
run c4_128
run c4_256            --instances 256
run c4_512_rounds     --instances 512
run c4_512_packed     --instances 512 --opt pack_roots=1 --opt pack_max=4
run c4_1024_rounds    --instances 1024
run c4_1024_packed_k2 --instances 1024 --opt pack_roots=1 --opt pack_islands=2
run c4_1024_packed    --instances 1024 --opt pack_roots=1 --opt pack_max=4
