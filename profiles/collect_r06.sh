#!/bin/bash
# Run ON THE GPU BOX (gpurun): for each named command (1) the command alone (its own bench line: the step time WITHOUT a profiler),
# (2) a kernel-trace/stats pass and (3, 4) — in separate passes, never mixed with trace domains (MI355X_MICROARCH.md) — the HBM
# counters FETCH_SIZE / WRITE_SIZE. profiles/summarize_cfg.py condenses them into <name>_n1_rocprof_summary.json, which also records
# the step time the command reported in passes (1) and (2): the kernel trace's durations are only evidence for the bench line if the
# profiled run's own step time agrees with the unprofiled one (C4's did not).
# usage: bash profiles/collect_r06.sh <tag> <name>...      names: c2 c3 c4 c4x64 c4x256 c1 c5 taps
cd "${GRAFT_REPO_ROOT:-/root/repo}"; R=$PWD; export TMPDIR=/tmp
TAG=${1:-r06}; shift; O=$R/gpurun_out/$TAG; mkdir -p $O
for name in "$@"; do
  case $name in
    c2) CMD="python $R/bench.py --no-cpu-baseline --no-configs --steps 20 --warmup 5";;
    c3) CMD="python $R/benchmarks/driver_configs.py c3 --gpu-only";;
    c4) CMD="python $R/bench.py --workload c4 --no-cpu-baseline --no-host-leg --no-full-chip --steps 12 --warmup 3";;
    c4x64) CMD="python $R/bench.py --workload c4 --instances 64 --no-cpu-baseline --no-host-leg --no-full-chip --steps 12 --warmup 3";;
    c4x256) CMD="python $R/bench.py --workload c4 --instances 256 --no-cpu-baseline --no-host-leg --no-full-chip --steps 12 --warmup 3";;
    c1) python -m elementary_amd.tools dump c1 $O/c1_batch.json > /dev/null 2>&1; CMD="$R/examples/bench_cli $O/c1_batch.json 4000 44100";;   # (the native host itself: rocprofv3 does not follow a subprocess)
    c5) CMD="python $R/benchmarks/driver_configs.py c5 --gpu-only";;
    taps) CMD="python $R/benchmarks/driver_configs.py taps --gpu-only";;
    *) echo "unknown $name"; continue;;
  esac
  echo "$CMD" > $O/prof_${name}_cmd.txt
  (cd /tmp && timeout 200 $CMD < /dev/null > $O/prof_${name}_plain.log 2>&1)
  (cd /tmp && timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_${name}_stats -- $CMD < /dev/null > $O/prof_${name}_stats.log 2>&1)
  (cd /tmp && timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/prof_${name}_fetch -- $CMD < /dev/null > $O/prof_${name}_fetch.log 2>&1)
  (cd /tmp && timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/prof_${name}_write -- $CMD < /dev/null > $O/prof_${name}_write.log 2>&1)
  timeout 60 python $R/profiles/summarize_cfg.py $O $name $O/${name}_n1_rocprof_summary.json "$CMD" < /dev/null
  # keep the merge-back small: the per-dispatch CSVs stay on the box (the stats CSVs are small and come back)
  find $O -name "*counter_collection.csv" -size +4M -delete; find $O -name "*kernel_trace.csv" -size +4M -delete
done
