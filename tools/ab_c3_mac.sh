cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/r06e; mkdir -p $O
for m in ${MODES:-0 3 0 3}; do
  ELEMHIP_C3_OPTS=conv_long_mac_lds=$m timeout 120 python benchmarks/driver_configs.py c3 --gpu-only 2>/dev/null | tail -n 1 > $O/c3_mac_mode_${m}_$RANDOM.json
done
for f in $O/c3_mac_mode_*.json; do echo "$f $(python -c "import json,sys; d=json.load(open('$f')); print(round(d['us_per_block'],5), d['launch_us_per_step'])")"; done

