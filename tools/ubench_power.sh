#!/bin/bash
# tools/ubench_power.sh -- run each instruction mix of ubench_power for a few seconds and sample clock / power meanwhile
R=${GRAFT_REPO_ROOT:-/root/repo}
for k in mad mad_sgpr mad29 mix alignbit add lds; do
  $R/tools/ubench_power $k 4 > /tmp/up_$k.txt &
  BP=$!
  sleep 1.5
  for i in 1 2 3 4; do
    rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Socket" | sed -e 's/.*sclk clock level: //' -e 's/.*Power (W)://' | tr '\n' ' '
    echo
    sleep 0.5
  done | tr '\n' ';'
  echo
  wait $BP
  cat /tmp/up_$k.txt
done
