#!/bin/bash
# C3 bring-up of the matrix-core partition MAC (conv.hip): parity in both modes, timings per mode, per-kernel profile.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; R=$PWD
O=$R/gpurun_out/${1:-r4h}; mkdir -p $O; export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_convolve.py -q -x -k "matrix_cores or wasm_recording or large_launch or interleave" < /dev/null 2>&1 | tail -4 | tee $O/pytest_conv.txt
for m in 1 0; do timeout 120 python benchmarks/bench_configs.py c3 --opt conv_mfma=$m < /dev/null > $O/c3_mode$m.json 2> $O/c3_mode$m.err; timeout 20 python -c "
import json; d=json.loads(open('$O/c3_mode$m.json').read().strip().splitlines()[-1]); print('mode $m', d['gpu_us_per_block'], d['gpu_launch_set_profile'])" < /dev/null; done
for m in 1 0; do (cd /tmp && timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c3_m$m -- python $R/benchmarks/bench_configs.py c3 --opt conv_mfma=$m < /dev/null > $O/prof_c3_m$m.log 2>&1); f=$(find $O/prof_c3_m$m -name "*kernel_stats.csv" 2>/dev/null | head -1); echo "mode $m stats: $f"; [ -n "$f" ] && head -7 "$f" | cut -d, -f1-4 | cut -c1-110; done
