#!/bin/bash
# ONE parametrised GPU-box job script (replaces the per-run scratch scripts of earlier rounds):
#   tools/gpu_job.sh <out-tag> <step> [<step> ...]     steps: tests | tests:<pytest args> | bench | bench:<args> | c4 | cfg:<name> | prof:<cfgs> | py:<script args>
# Everything lands under gpurun_out/<out-tag>/ ; each step prints a short tail so the gpurun verdict shows what happened.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; R=$PWD; TAG=${1:-job}; shift; O=$R/gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp
for step in "$@"; do
  kind=${step%%:*}; arg=""; [[ "$step" == *:* ]] && arg=${step#*:}
  echo "=== $step"
  case $kind in
    tests) timeout 1500 python -m pytest ${arg:-tests} -m gpu -x -q < /dev/null > $O/pytest_$(echo "$arg" | tr -c 'A-Za-z0-9\n' _ | cut -c1-40).log 2>&1; echo "pytest rc=$?"; tail -n 8 $O/pytest_*.log | cut -c1-400;;
    bench) timeout 600 python bench.py ${arg:---steps 20 --warmup 5} < /dev/null > $O/bench.out 2> $O/bench.err; echo "rc=$?"; tail -n 1 $O/bench.out | wc -c; tail -n 1 $O/bench.out | cut -c1-3800;;
    c4)    timeout 600 python bench.py --workload c4 $arg < /dev/null > $O/bench_c4.out 2> $O/bench_c4.err; echo "rc=$?"; tail -n 1 $O/bench_c4.out | cut -c1-2500;;
    cfg)   timeout 400 python benchmarks/driver_configs.py $arg < /dev/null > $O/cfg_$(echo "$arg" | tr -c 'A-Za-z0-9\n' _).json 2> $O/cfg_$(echo "$arg" | tr -c 'A-Za-z0-9\n' _).err; echo "rc=$?"; tail -n 1 $O/cfg_$(echo "$arg" | tr -c 'A-Za-z0-9\n' _).json | cut -c1-3000;;
    prof)  timeout 1500 bash profiles/collect_r06.sh $TAG $arg < /dev/null 2>&1 | tail -n 40;;
    py)    timeout 900 python $arg < /dev/null > $O/py_$(echo "$arg" | tr -c 'A-Za-z0-9\n' _ | cut -c1-50).out 2>&1; echo "rc=$?"; tail -n 25 $O/py_$(echo "$arg" | tr -c 'A-Za-z0-9\n' _ | cut -c1-50).out | cut -c1-600;;
    sh)    timeout 900 bash -c "$arg" < /dev/null 2>&1 | tail -n 40;;
    *) echo "unknown step $step";;
  esac
done
