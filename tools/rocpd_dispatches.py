#!/usr/bin/env python3
"""tools/rocpd_dispatches.py DB [substr] -- per-dispatch durations (us), the gap since the previous dispatch ended, and grids
of the kernels whose name contains substr, in launch order, from a rocprofv3 --kernel-trace rocpd database."""
import sqlite3
import sys

con = sqlite3.connect(sys.argv[1])
cur = con.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
sub = sys.argv[2] if len(sys.argv) > 2 else ""
want = [c for c in ("name", "start", "end", "grid_x", "grid_y", "grid_size_x", "grid_size_y", "workgroup_size_x", "lds_size", "lds_block_size") if c in cols]
rows = cur.execute("select %s from kernels order by start" % ", ".join(want)).fetchall()
prev_end = None
for r in rows:
    d = dict(zip(want, r))
    gap = (d["start"] - prev_end) / 1e3 if prev_end is not None else 0.0
    prev_end = d["end"]
    if sub in d["name"]:
        nm = d["name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "").replace("lcpc::", "")[:48]
        print("%-48s %9.1f us  (gap %7.1f)  grid %s x %s" % (nm, (d["end"] - d["start"]) / 1e3, gap, d.get("grid_x", d.get("grid_size_x")), d.get("grid_y", d.get("grid_size_y"))))
