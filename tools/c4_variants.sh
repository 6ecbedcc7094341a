#!/bin/bash
# NOTE (r05): ELEMHIP_JIT_DEFINES and the ELEMHIP_EXP_* / *_CHAINS hooks exist only in a library built with `make -C elementary_amd/csrc EXPERIMENTAL=1`.
# Run ON THE GPU BOX: C4 (128 independent render instances) kernel time and per-wave busy trace for JIT-define / option variants.
# usage: tools/c4_variants.sh outdir "DEFINES|opt1=v opt2=v" ...
out=$1; shift; mkdir -p $out
i=0
for v in "$@"; do
  i=$((i+1))
  defs="${v%%|*}"; opts="${v#*|}"; [ "$opts" = "$v" ] && opts=""
  echo "== variant $i: defines [$defs] options [$opts]" | tee -a $out/summary.txt
  ELEMHIP_JIT_DEFINES="$defs" ELEMHIP_TRACE_OPTS="$opts" timeout 200 python tools/spec_trace.py c4 64 2>/dev/null | grep "^wave\|blocks_rendered" | sed 's/start.*total/total/' | cut -c1-260 | tee -a $out/summary.txt
done
