#!/bin/bash
# C3 bring-up of the matrix-core partition MAC (conv.hip): layout probe, parity in both modes, timings per mode.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/${1:-r4f}; mkdir -p $O
timeout 30 tools/micro/mfma4x4_layout_bin | tee $O/mfma4x4_layout.txt
timeout 400 python -m pytest tests/test_gpu_convolve.py -q -x -k "matrix_cores or wasm_recording or large_launch" 2>&1 | tail -5 | tee $O/pytest_conv.txt
for m in 1 0 2; do timeout 200 python benchmarks/bench_configs.py c3 --opt conv_mfma=$m > $O/c3_mode$m.json 2> $O/c3_mode$m.err; python -c "
import json; d=json.loads(open('$O/c3_mode$m.json').read().strip().splitlines()[-1]); print('mode $m', d['gpu_us_per_block'], d['gpu_launch_set_profile'])"; done
