#!/bin/bash
# Run ON THE GPU BOX (gpurun): kernel-trace stats and, in separate passes (MI355X_MICROARCH.md: FETCH_SIZE and
# WRITE_SIZE do not fit one pass; never mixed with trace domains), the HBM counters for the bench.py command.
# Outputs land in gpurun_out/prof_* ; profiles/summarize.py condenses them into profiles/<round>/.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
REPO=$PWD
BENCH_ARGS="${BENCH_ARGS:---no-cpu-baseline --steps 20 --warmup 5}"
CMD="python $REPO/bench.py $BENCH_ARGS"
OUT=$REPO/gpurun_out
mkdir -p $OUT
echo "python bench.py $BENCH_ARGS" > $OUT/prof_cmd.txt
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_stats -- $CMD > $OUT/prof_stats.log 2>&1)
(cd /tmp && rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/prof_fetch -- $CMD > $OUT/prof_fetch.log 2>&1)
(cd /tmp && rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/prof_write -- $CMD > $OUT/prof_write.log 2>&1)
# BASELINE configs[2] (C3) through the multi-block convolve kernels: per-kernel durations only
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_c3 -- python $REPO/benchmarks/bench_configs.py c3 > $OUT/prof_c3.log 2>&1)
tail -1 $OUT/prof_stats.log | cut -c1-300
find $OUT/prof_stats $OUT/prof_fetch $OUT/prof_write -name "*.csv" | head -20
python profiles/summarize.py $OUT ${ROUND_TAG:-r03}
