#!/bin/bash
# tools/prof_modes.sh -- kernel-trace the headline bench with canonical comm (default) and with LCPC_COMM_MONT=1;
# prints the per-kernel summary and the durations of the NTT launches in order (pass 1, pass 2, ...).  Run via gpurun.
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
for mode in canon mont; do
  if [ $mode = mont ]; then export LCPC_COMM_MONT=1; else unset LCPC_COMM_MONT; fi
  rm -rf $R/gpurun_out/kt_$mode
  rocprofv3 --kernel-trace --stats -d $R/gpurun_out/kt_$mode -o bench -- python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline > $R/gpurun_out/kt_$mode.log 2>&1
  DB=$(find $R/gpurun_out/kt_$mode -name '*.db' | head -1)
  python $R/tools/rocpd_summary.py $DB > $R/gpurun_out/kt_$mode.txt 2>&1
  echo "== $mode"; head -6 $R/gpurun_out/kt_$mode.txt
  python - "$DB" <<'PY'
import sqlite3, sys
con = sqlite3.connect(sys.argv[1])
rows = con.execute("select name, start, end from kernels order by start").fetchall()
d = [(e - s) / 1e3 for n, s, e in rows if "ntt_pass" in n]
print("ntt launches (us):", " ".join("%.0f" % x for x in d[-10:]))
PY
done
