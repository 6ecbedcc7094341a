#!/bin/bash
# one-off counter probe of a command: tools/pmc_probe.sh <out-tag> "<counters>" <command...>   (each counter set in its own pass, no trace domains)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; R=$PWD; export TMPDIR=/tmp; TAG=$1; CTRS=$2; shift 2; O=$R/gpurun_out/$TAG; mkdir -p $O
(cd /tmp && timeout 300 rocprofv3 --pmc $CTRS --output-format csv -d $O/pmc -- "$@" < /dev/null > $O/pmc.log 2>&1)
f=$(find $O/pmc -name "*counter_collection.csv" | head -1)
python - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    k = r["Kernel_Name"].split("(")[0][-48:]
    agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    if "elemhip" not in k: continue
    print(k, {c: round(sum(v) / len(v)) for c, v in d.items()}, "n=", len(next(iter(d.values()))))
PY
