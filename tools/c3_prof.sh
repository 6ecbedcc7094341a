cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/r4g; mkdir -p $O; export TMPDIR=/tmp
timeout 60 tools/micro/mfma4x4_rate_bin | tee $O/mfma4x4_rate.txt
for m in 1 0; do (cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c3_m$m -- python $GRAFT_REPO_ROOT/benchmarks/bench_configs.py c3 --opt conv_mfma=$m > $O/prof_c3_m$m.log 2>&1); f=$(find $O/prof_c3_m$m -name "*kernel_stats.csv" | head -1); echo "mode $m"; head -8 $f | cut -d, -f1-4 | cut -c1-120; done
