#!/usr/bin/env python3
"""tools/pmc_to_json.py FETCH.db WRITE.db OUT.json [SQ.db] -- per-kernel average FETCH_SIZE / WRITE_SIZE (KB per dispatch)
(and SQ_INSTS_VALU wave-instructions per dispatch when the SQ pass is given) from separate rocprofv3 --pmc passes of `python bench.py ...`, for bench.py's roofline.traffic field.  The file is stamped
with the hash of the device sources it was measured on (bench.kernel_stamp): bench.py quotes it only while that matches."""
import json
import os
import re
import sqlite3
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench


def avg(db, counter):
    cur = sqlite3.connect(db).cursor()
    agg = {}
    for name, val in cur.execute("select kernel_name, value from counters_collection where counter_name = ?", (counter,)):
        k = re.sub(r"\(.*", "", name.replace("(anonymous namespace)::", "")).replace("void ", "").replace("lcpc::", "")
        if k.startswith("at::") or k.startswith("__amd"):
            continue
        agg.setdefault(k, []).append(float(val))
    return {k: sum(v) / len(v) for k, v in agg.items()}


f, w = avg(sys.argv[1], "FETCH_SIZE"), avg(sys.argv[2], "WRITE_SIZE")
v = avg(sys.argv[4], "SQ_INSTS_VALU") if len(sys.argv) > 4 else {}
out = {"kernel_stamp": bench.kernel_stamp(), "borrow_coeffs": os.environ.get("LCPC_PMC_BORROW") is not None,   # profile_round.sh profiles bench.py in its default (owning) mode
       "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python bench.py --steps 2 --warmup 1 --lean",
       "note": "KB per dispatch; on gfx950 FETCH_SIZE counts half the bytes of a wide coalesced read (MI355X_MICROARCH.md): "
               "hbm_bytes = (2*FETCH_KB + WRITE_KB) * 1024",
       "kernels": {k: dict({"FETCH_KB": round(f.get(k, 0.0), 1), "WRITE_KB": round(w.get(k, 0.0), 1)},
                            **({"SQ_INSTS_VALU": round(v[k])} if k in v else {})) for k in sorted(set(f) | set(w))}}
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps(out["kernels"], indent=1))
