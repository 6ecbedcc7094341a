#!/bin/bash
# NOTE (r05): ELEMHIP_JIT_DEFINES and the ELEMHIP_EXP_* / *_CHAINS hooks exist only in a library built with `make -C elementary_amd/csrc EXPERIMENTAL=1`.
# Run ON THE GPU BOX: bench + per-wave busy trace of the C2 voice kernel for a list of JIT-define / option variants.
# usage: tools/ab_variants.sh outdir "DEFINES|opt1=v opt2=v" ...
out=$1; shift; mkdir -p $out
i=0
for v in "$@"; do
  i=$((i+1))
  defs="${v%%|*}"; opts="${v#*|}"; [ "$opts" = "$v" ] && opts=""
  optargs=""; for o in $opts; do optargs="$optargs --opt $o"; done
  echo "== variant $i: defines [$defs] options [$opts]" | tee -a $out/summary.txt
  ELEMHIP_JIT_DEFINES="$defs" timeout 200 python bench.py --steps 20 --warmup 3 --batch-blocks 256 --no-cpu-baseline --device-resident $optargs > $out/bench_$i.json 2> $out/bench_$i.err
  python - $out/bench_$i.json <<'PY' | tee -a $out/summary.txt
import json,sys
try:
    b=json.load(open(sys.argv[1])); print("  value %.2f M  us/block %.3f  levels %s" % (b["value"]/1e6, b["us_per_block"], [round(x,1) for x in b["roofline"]["launch_us_per_step"]]))
except Exception as e: print("  bench failed", e)
PY
  ELEMHIP_JIT_DEFINES="$defs" ELEMHIP_TRACE_OPTS="$opts" timeout 120 python tools/spec_trace.py c2 64 2>/dev/null | grep "^wave" | sed 's/start.*total/total/' | tee -a $out/summary.txt
done
