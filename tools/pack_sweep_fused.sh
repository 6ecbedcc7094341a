#!/bin/bash
# Run ON THE GPU BOX: lane-packing with and without the svf coefficient pre-pass inside the scan (`fuse_svf_coef`), C2 at 512 / 1024
# voices per GPU, device-resident 128-block launch sets.
out=$1; mkdir -p $out
run() { tag=$1; shift; timeout 300 python bench.py --no-cpu-baseline --device-resident --steps 8 --warmup 2 "$@" > $out/$tag.json 2> $out/$tag.err
  python - $out/$tag.json $tag <<'PY' | tee -a $out/summary.txt
import json,sys
try:
    b=json.load(open(sys.argv[1])); c=b["config"]
    print("%-26s value %9.2f M  us/block %8.3f  islands %s  K %s  copies %s" % (sys.argv[2], b["value"]/1e6, b["us_per_block"], c.get("islands"), c.get("voices_per_island","-"), c.get("pipelined_blocks_in_flight","-")))
except Exception as e: print(sys.argv[2], "failed", e)
PY
}
run c2_256_plain          --batch-blocks 128
run c2_256_fused          --batch-blocks 128 --opt fuse_svf_coef=1
for v in 512 1024; do
  run c2_${v}_nopack        --batch-blocks 128 --voices $v --opt pack_islands=1
  run c2_${v}_k2_plain      --batch-blocks 128 --voices $v --opt pack_islands=2 --opt fuse_svf_coef=0
  run c2_${v}_k2_fused      --batch-blocks 128 --voices $v --opt pack_islands=2 --opt fuse_svf_coef=1
done
run c2_1024_k3_fused      --batch-blocks 128 --voices 1024 --opt pack_islands=3 --opt pack_max=4 --opt fuse_svf_coef=1
run c2_1024_k4_fused      --batch-blocks 128 --voices 1024 --opt pack_islands=4 --opt pack_max=4 --opt fuse_svf_coef=1
run c2_1024_k4_plain      --batch-blocks 128 --voices 1024 --opt pack_islands=4 --opt pack_max=4 --opt fuse_svf_coef=0
