#!/usr/bin/env python3
"""tools/spike_sdig_rows.py -- how does the Brakedown SpMM time scale with the number of rows at the C3 row length
(166 292 -> 252 931)?  The gathered operand of one matrix entry is n_rows x 32 B; the working set of the first precode
level is 166 292 x n_rows x 32 B (537 MB at 101 rows, more than the 256 MB Infinity Cache).  Run under
rocprofv3 --kernel-trace and compare spmm_t_kernel time per row."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tools")]
import json
import time

import torch

import bench_configs as B
from lcpc_amd import LcCommit, SdigEncoding

npr = 166292
enc = SdigEncoding.new_from_dims(3, npr, 252931, 0, 3)
for rows in (16, 26, 51, 101, 202, 404):
    n = rows * npr
    coeffs = B.rand_coeffs(n, 4, 1)
    st = torch.cuda.current_stream().cuda_stream
    c = LcCommit(enc)
    for _ in range(2):
        LcCommit.commit_device(coeffs.data_ptr(), n, enc, st, sync=False, into=c)
    torch.cuda.synchronize()
    c.set_timing(True)
    LcCommit.commit_device(coeffs.data_ptr(), n, enc, st, sync=True, into=c)
    tm = c.timings()
    c.set_timing(False)
    print(json.dumps({"rows": rows, "encode_ms": round(tm.encode_ms, 3), "encode_us_per_row": round(tm.encode_ms * 1e3 / rows, 2),
                      "hash_ms": round(tm.hash_ms, 3)}), flush=True)
    del coeffs
