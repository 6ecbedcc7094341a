#!/bin/bash
# C4 biquad instruction forms A/B (engine option biquad_form -> ELEMHIP_BIQUAD_FORM of the run-time compiled island kernels), each with
# the bench line's own parity check on: usage: bash tools/c4_forms.sh <out-tag> <form>...
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/${1:-c4forms}; shift; mkdir -p $O
for f in "$@"; do
  timeout 300 python bench.py --workload c4 --opt biquad_form=$f --no-cpu-baseline --steps 12 --warmup 3 < /dev/null > $O/c4_form$f.out 2> $O/c4_form$f.err
  tail -n 1 $O/c4_form$f.out | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('form $f', round(d['value']/1e9,3), 'G/s', round(d['ms_per_step'],3), 'ms/step', d['roofline'].get('launch_us_per_step'), 'parity', d['parity'])"
done
