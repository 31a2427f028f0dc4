#!/bin/bash
# tools/sample_clocks.sh -- sample sclk / power with rocm-smi while the headline bench loops (is the NTT clock- or power-limited?)
R=${GRAFT_REPO_ROOT:-/root/repo}
python $R/bench.py --steps 1500 --warmup 2 --no-cpu-baseline > $R/gpurun_out/clk_bench.log 2>&1 &
BP=$!
while kill -0 $BP 2>/dev/null; do
  rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Socket" | sed -e 's/.*sclk clock level: //' -e 's/.*Power (W)://' | tr '\n' ' '
  echo
  sleep 0.5
done | grep -v "(9[0-9]Mhz)\|(1[0-9][0-9]Mhz)" | tail -25
tail -1 $R/gpurun_out/clk_bench.log | cut -c1-200
