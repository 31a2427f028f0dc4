#!/bin/bash
# tools/trace_small.sh -- launch-by-launch kernel times of small commitments (the last commit of five each): C1 (Ft63 2^16),
# Ft127 2^16, Ft255 2^13 / 2^15 / 2^17 / 2^19.  Run on the MI355X box: gpurun -- 'bash tools/trace_small.sh > gpurun_out/small_trace.txt'
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
for args in "0 1 16" "1 2 16" "3 4 13" "3 4 15" "3 4 17" "3 4 19"; do
  rm -rf /tmp/ts; rocprofv3 --kernel-trace -d /tmp/ts -o t -- python $R/tools/trace_small_any.py $args > /dev/null 2>&1
  echo "== field / limbs / log2(coefficients): $args"
  python $R/tools/rocpd_dispatches.py $(find /tmp/ts -name '*.db' | head -1) | tail -7
done
