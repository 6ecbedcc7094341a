#!/bin/bash
# Run ON THE GPU BOX: lane-packing sweep — C2 at 256 / 512 / 1024 voices and C4 at 128 / 512 / 1024 instances per GPU, packing off / auto / forced.
out=$1; mkdir -p $out
run() { tag=$1; shift; timeout 300 python bench.py --no-cpu-baseline --steps 8 --warmup 2 "$@" > $out/$tag.json 2> $out/$tag.err
  python - $out/$tag.json $tag <<'PY' | tee -a $out/summary.txt
import json,sys
try:
    b=json.load(open(sys.argv[1])); c=b["config"]
    per=b.get("us_per_block", b.get("us_per_block_step"))
    print("%-22s value %9.2f M  us/block %8.3f  islands %s  K %s  copies %s" % (sys.argv[2], b["value"]/1e6, per, c.get("islands"), c.get("voices_per_island","-"), c.get("pipelined_blocks_in_flight","-")))
except Exception as e: print(sys.argv[2], "failed", e)
PY
}
run c2_256            --batch-blocks 256
run c2_512_nopack     --batch-blocks 256 --voices 512 --opt pack_islands=1
run c2_512_auto       --batch-blocks 256 --voices 512
run c2_1024_nopack    --batch-blocks 128 --voices 1024 --opt pack_islands=1
run c2_1024_k2        --batch-blocks 128 --voices 1024 --opt pack_islands=2
run c2_1024_auto      --batch-blocks 128 --voices 1024
run c2_1024_k4d1      --batch-blocks 128 --voices 1024 --opt pack_islands=4 --opt pipeline_copies=1
run c4_128            --workload c4 --batch-blocks 64 --instances 128
run c4_512_nopack     --workload c4 --batch-blocks 64 --instances 512 --opt pack_islands=1
run c4_512_auto       --workload c4 --batch-blocks 64 --instances 512
run c4_1024_nopack    --workload c4 --batch-blocks 64 --instances 1024 --opt pack_islands=1
run c4_1024_auto      --workload c4 --batch-blocks 64 --instances 1024
run c4_128_k2         --workload c4 --batch-blocks 64 --instances 128 --opt pack_islands=2
