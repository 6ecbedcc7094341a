#!/bin/bash
# Run ON THE GPU BOX (gpurun): for each named command a kernel-trace/stats pass and, in separate passes (MI355X_MICROARCH.md:
# FETCH_SIZE and WRITE_SIZE do not fit one pass; never mixed with trace domains), the HBM counters. Summaries land in
# gpurun_out/<tag>/ and are copied into profiles/<round>/ from the authoring container.
# usage: bash profiles/collect_r04.sh <tag> <name>...      names: c2 c3 c4
cd "${GRAFT_REPO_ROOT:-/root/repo}"; R=$PWD; export TMPDIR=/tmp
TAG=${1:-r04}; shift; O=$R/gpurun_out/$TAG; mkdir -p $O
for name in "$@"; do
  case $name in
    c2) CMD="python $R/bench.py --no-cpu-baseline --steps 20 --warmup 5";;
    c3) CMD="python $R/benchmarks/bench_configs.py c3";;
    c4) CMD="python $R/bench.py --workload c4 --no-cpu-baseline --steps 12 --warmup 3";;
    *) echo "unknown $name"; continue;;
  esac
  echo "$CMD" > $O/prof_${name}_cmd.txt
  (cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_${name}_stats -- $CMD < /dev/null > $O/prof_${name}_stats.log 2>&1)
  (cd /tmp && timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/prof_${name}_fetch -- $CMD < /dev/null > $O/prof_${name}_fetch.log 2>&1)
  (cd /tmp && timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/prof_${name}_write -- $CMD < /dev/null > $O/prof_${name}_write.log 2>&1)
  timeout 60 python $R/profiles/summarize_cfg.py $O $name $O/${name}_n1_rocprof_summary.json "$CMD" < /dev/null
  # keep the merge-back small: the per-dispatch CSVs stay on the box
  find $O -name "*counter_collection.csv" -size +4M -delete; find $O -name "*kernel_trace.csv" -size +4M -delete
done
