cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/r06e; mkdir -p $O
for m in 0 2 1 0 2; do
  ELEMHIP_C3_OPTS=conv_long_mac_lds=$m timeout 120 python benchmarks/driver_configs.py c3 --gpu-only 2>/dev/null | tail -n 1 > $O/c3_mac_mode_${m}_$RANDOM.json
done
for f in $O/c3_mac_mode_*.json; do echo "$f $(python -c "import json,sys; d=json.load(open('$f')); print(round(d['us_per_block'],5), d['launch_us_per_step'])")"; done
timeout 300 python tools/c5_commit_breakdown.py 100 > $O/c5_breakdown.json 2>$O/c5_breakdown.err; cut -c1-3000 $O/c5_breakdown.json
