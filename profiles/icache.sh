#!/bin/bash
# Run ON THE GPU BOX: instruction-cache and issue counters of the island kernel for the bench.py command (own pmc passes).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
REPO=$PWD
CMD="python $REPO/bench.py --no-cpu-baseline --steps 512 --warmup 64"
OUT=$REPO/gpurun_out
mkdir -p $OUT
(cd /tmp && rocprofv3 -L 2>/dev/null | grep -i -E "icache|SQ_INSTS|SQ_WAVE_CYCLES|SQ_BUSY_CY|SQ_INST_CYCLES|IFETCH|SQ_WAIT_INST|SQ_ACTIVE_INST" | cut -c1-160 | sort -u | head -60) > $OUT/counters_list.txt
i=0
for set in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_IFETCH SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM"; do
  i=$((i+1))
  (cd /tmp && timeout 300 rocprofv3 --pmc $set --output-format csv -d $OUT/prof_ic$i -- $CMD > $OUT/prof_ic$i.log 2>&1)
  f=$(find $OUT/prof_ic$i -name "*counter_collection.csv" | head -1)
  echo "== $set -> $f"
  [ -n "$f" ] && python - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: [0.0, 0])
for r in csv.DictReader(open(sys.argv[1])):
    k = (r["Kernel_Name"][:40], r["Counter_Name"])
    acc[k][0] += float(r["Counter_Value"]); acc[k][1] += 1
for (k, c), (v, n) in sorted(acc.items()):
    if "island" in k: print(f"{k:40s} {c:24s} sum {v:.4g}  per-dispatch {v/n:.4g}  n {n}")
PY
done
