#!/bin/bash
# NOTE (r05): ELEMHIP_JIT_DEFINES and the ELEMHIP_EXP_* / *_CHAINS hooks exist only in a library built with `make -C elementary_amd/csrc EXPERIMENTAL=1`.
# Cost of polling without s_wakeup (island_ops.inc ELEMHIP_WAKE): the C2 and C4 benches for several poll periods, and the old
# ping-and-sleep-8 protocol for comparison (measurement only: it is the one that corrupts other waves' wait states).
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/${1:-r4e}; mkdir -p $O
for v in "ELEMHIP_SPEC_SLEEP=1" "ELEMHIP_SPEC_SLEEP=2" "ELEMHIP_SPEC_SLEEP=0" "ELEMHIP_SPEC_SLEEP=4" "ELEMHIP_EXP_WAKEUP=1 ELEMHIP_SPEC_SLEEP=8"; do
  tag=$(echo "$v" | tr ' =' '__')
  ELEMHIP_JIT_DEFINES="$v" timeout 100 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --device-resident > $O/c2_$tag.json 2> $O/c2_$tag.err
  echo "c2 $v: $(python -c "import json,sys; d=json.loads(open('$O/c2_$tag.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], d.get('parity'))" 2>&1 | tail -1)"
  ELEMHIP_JIT_DEFINES="$v" timeout 100 python bench.py --workload c4 --steps 4 --warmup 1 --no-cpu-baseline --device-resident > $O/c4_$tag.json 2> $O/c4_$tag.err
  echo "c4 $v: $(python -c "import json,sys; d=json.loads(open('$O/c4_$tag.json').read().strip().splitlines()[-1]); print(d['us_per_block_step'])" 2>&1 | tail -1)"
done
