#!/usr/bin/env python3
"""tools/ab_c3_rowgroups.py -- Brakedown C3 (2^24 Ft255 coefficients: 101 rows x 166292 -> 252931) with the wide levels encoded in
row groups (LCPC_SDIG_ROW_GROUP, read at context creation): one launch of the packed (output, row) kernel per group of <= g rows
over all outputs, so that one group's gather range (166292 x g x 32 B at level 0) can sit in the 256 MiB Infinity Cache.
A/B in ONE process, variants interleaved round-robin; one JSON line per variant (profiles/r04_c3_rowgroups.jsonl).  Roots of all
variants must agree (the parity tests pin the default path to the oracle)."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tools")]
import torch

import bench_configs as B
from lcpc_amd import LcCommit, SdigEncoding

n = 1 << 24
# an integer = a row-group size (0 = the default path); "p1" / "p2" = the pricing experiment LCPC_DEBUG_K2_PRICE=1 / 2 (timing
# only: what a limb-form T would cost in the wide levels -- no packed -> 29-bit conversion of the gathered operand / that plus the
# ninth limb's bytes; roots are wrong by construction and not compared)
groups = [x if x.startswith("p") else int(x) for x in (sys.argv[1:] or ["0", "51", "34", "26", "21", "17"])]
coeffs = B.rand_coeffs(n, 4, 1)
st = torch.cuda.current_stream().cuda_stream
var = {}
for g in groups:
    if isinstance(g, str):
        os.environ["LCPC_DEBUG_K2_PRICE"] = g[1:]
    elif g:
        os.environ["LCPC_SDIG_ROW_GROUP"] = str(g)
    try:
        enc = SdigEncoding.new(3, n, 0)
    finally:
        os.environ.pop("LCPC_SDIG_ROW_GROUP", None)
        os.environ.pop("LCPC_DEBUG_K2_PRICE", None)
    c = LcCommit(enc)
    for _ in range(3):
        LcCommit.commit_device(coeffs.data_ptr(), n, enc, st, sync=False, borrow=True, into=c)
    torch.cuda.synchronize()
    var[g] = {"enc": enc, "c": c, "ms": [], "root": c.get_root()}
assert len({v["root"] for g, v in var.items() if not isinstance(g, str)}) == 1, "row-group variants disagree"
for rnd in range(8):
    for g in groups:
        v = var[g]
        for _ in range(3):
            LcCommit.commit_device(coeffs.data_ptr(), n, v["enc"], st, sync=False, borrow=True, into=v["c"])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            LcCommit.commit_device(coeffs.data_ptr(), n, v["enc"], st, sync=False, borrow=True, into=v["c"])
        torch.cuda.synchronize()
        v["ms"].append((time.perf_counter() - t0) / 20 * 1e3)
for g in groups:
    v = var[g]
    c = v["c"]
    enc_ms = []
    c.set_timing(True)
    for _ in range(5):
        LcCommit.commit_device(coeffs.data_ptr(), n, v["enc"], st, sync=True, borrow=True, into=c)
        enc_ms.append(c.timings().encode_ms)
    c.set_timing(False)
    rows = c.n_rows
    if isinstance(g, str):
        print(json.dumps({"variant": "LCPC_DEBUG_K2_PRICE=" + g[1:], "commit_ms_mean": round(sum(v["ms"]) / len(v["ms"]), 3),
                          "commit_ms_min": round(min(v["ms"]), 3), "encode_ms_min": round(min(enc_ms), 3)}), flush=True)
        continue
    G = (rows + g - 1) // g if g else 1
    print(json.dumps({"row_group": g, "groups": G, "rows_per_group": (rows + G - 1) // G if g else rows,
                      "level0_gather_MB_per_group": round(166292 * ((rows + G - 1) // G if g else rows) * 32 / 1e6, 1),
                      "commit_ms_mean": round(sum(v["ms"]) / len(v["ms"]), 3), "commit_ms_min": round(min(v["ms"]), 3),
                      "encode_ms_min": round(min(enc_ms), 3), "encode_ms_mean": round(sum(enc_ms) / len(enc_ms), 3)}), flush=True)
