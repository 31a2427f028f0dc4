#!/usr/bin/env python3
"""tools/ab_mid_shapes.py -- the two K1s passes with the packed intermediate (LCPC_NTT_MID_MAX_MB=0) against the 29-bit-limb
intermediate, on 2^26-element Ft255 commitments of different row lengths (first-pass runs of 2^LTJ elements: does the limb
format lose where its stores are partial lines?).  Prints encode_ms of an instrumented commit (mean of 5)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tools")]
import torch

import bench_configs as B
from lcpc_amd import LcCommit, LigeroEncoding

total = 26
for k in [int(x) for x in sys.argv[1:]] or (13, 15, 16, 17, 18, 19):
    n_cols, n_per_row = 1 << k, 1 << (k - 1)
    n = 1 << total
    coeffs = B.rand_coeffs(n, 4, 5)
    st = torch.cuda.current_stream().cuda_stream
    res = {}
    for mode in ("0", "6144", "0", "6144"):
        os.environ["LCPC_NTT_MID_MAX_MB"] = mode          # read once, when the encoder is created
        enc = LigeroEncoding.new_from_dims(3, n_per_row, n_cols)
        c = LcCommit(enc)
        for _ in range(3):
            LcCommit.commit_device(coeffs.data_ptr(), n, enc, st, sync=False, into=c)
        torch.cuda.synchronize()
        c.set_timing(True)
        ts = []
        for _ in range(5):
            LcCommit.commit_device(coeffs.data_ptr(), n, enc, st, sync=True, into=c)
            ts.append(c.timings().encode_ms)
        res.setdefault(mode, []).append(round(sum(ts) / len(ts), 3))
        del c
    print(json.dumps({"log_n_cols": k, "first_pass_stages": k - 10, "run_elems": 1 << (20 - k), "rows": n // n_per_row,
                      "encode_ms_packed": res["0"], "encode_ms_limbs": res["6144"]}), flush=True)
    del coeffs
