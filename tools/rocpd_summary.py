#!/usr/bin/env python3
"""tools/rocpd_summary.py -- turn rocprofv3's rocpd sqlite output (bench_results.db) into the plain-text
per-kernel summaries committed under profiles/ (kernel time stats; PMC counter sums/averages per kernel).

    python tools/rocpd_summary.py gpurun_out/prof/kt/bench_results.db            # --kernel-trace --stats
    python tools/rocpd_summary.py gpurun_out/prof/fetch/bench_results.db --pmc   # --pmc FETCH_SIZE ...
"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(.*", "", name.replace("(anonymous namespace)::", ""))
    name = name.replace("void ", "").replace("lcpc::", "")
    return name[:70]


def main():
    db = sys.argv[1]
    pmc = "--pmc" in sys.argv
    con = sqlite3.connect(db)
    cur = con.cursor()
    if not pmc:
        rows = cur.execute("select name, (end - start) from kernels").fetchall()
        agg = {}
        for n, d in rows:
            agg.setdefault(short(n), []).append(d)
        tot = sum(sum(v) for v in agg.values())
        print("%-72s %8s %12s %12s %12s %12s %7s" % ("KERNEL", "CALLS", "TOTAL_us", "AVG_us", "MIN_us", "MAX_us", "PCT"))
        for n, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
            print("%-72s %8d %12.1f %12.1f %12.1f %12.1f %6.2f%%" % (n, len(v), sum(v) / 1e3, sum(v) / len(v) / 1e3, min(v) / 1e3,
                                                                max(v) / 1e3, 100.0 * sum(v) / tot))
    else:
        cols = [r[1] for r in cur.execute("pragma table_info(counters_collection)")]
        print("# columns:", cols)
        rows = cur.execute("select kernel_name, counter_name, value from counters_collection").fetchall() \
            if "kernel_name" in cols else cur.execute("select * from counters_collection").fetchall()
        agg = {}
        for r in rows:
            k = (short(r[0]), r[1])
            agg.setdefault(k, []).append(float(r[2]))
        print("%-72s %-24s %8s %18s %18s" % ("KERNEL", "COUNTER", "DISPATCH", "SUM", "AVG_PER_DISPATCH"))
        for (n, c), v in sorted(agg.items()):
            print("%-72s %-24s %8d %18.1f %18.1f" % (n, c, len(v), sum(v), sum(v) / len(v)))


if __name__ == "__main__":
    main()
