#!/bin/bash
# dev probe (GPU box): shader clock while bench.py runs
cd "${GRAFT_REPO_ROOT:-/root/repo}"
rocm-smi --showperflevel 2>&1 | grep -i "perf" | head -3
rocm-smi --showclocks 2>&1 | grep -i -E "sclk|fclk|mclk" | head -6
(python bench.py --no-cpu-baseline --steps 1500000 --warmup 256 > gpurun_out/clk_bench.json 2>/dev/null) &
BP=$!
sleep 9
for i in 1 2 3 4; do rocm-smi --showclocks 2>&1 | grep -i -E "sclk" | head -2; rocm-smi --showpower 2>&1 | grep -i -E "power" | head -2; sleep 0.5; done
wait $BP
cut -c100-180 gpurun_out/clk_bench.json
