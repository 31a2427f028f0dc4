#!/usr/bin/env python3
"""tools/bench_c4_rank.py -- one rank's share of C4 (2^28 coefficients = 1024 rows x 262144 -> 524288 columns over 8 GPUs:
128 rows per GPU) on ONE GPU: the local commit of 128 rows at n_cols = 2^19, with the two NTT plans side by side
(two passes on 1024-element tiles through ntt_l9s.hip, or LCPC_NTT_GENERAL=1: the general kernel)."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from lcpc_amd import LcCommit, LigeroEncoding

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 128
for log_n in (19, 20):
    n_cols, npr = 1 << log_n, 1 << (log_n - 1)
    g = torch.Generator(device="cuda"); g.manual_seed(1)
    coeffs = torch.randint(-(1 << 63), (1 << 63) - 1, (rows * npr, 4), dtype=torch.int64, device="cuda", generator=g)
    coeffs[:, 3] &= (1 << 62) - 1
    roots = {}
    for plan in ("tile1024", "general"):
        if plan == "general":
            os.environ["LCPC_NTT_GENERAL"] = "1"
        try:
            enc = LigeroEncoding.new_from_dims(3, npr, n_cols)
        finally:
            os.environ.pop("LCPC_NTT_GENERAL", None)
        st = torch.cuda.current_stream().cuda_stream
        c = LcCommit(enc)
        for _ in range(3):
            LcCommit.commit_device(coeffs.data_ptr(), rows * npr, enc, st, sync=False, borrow=True, into=c)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            LcCommit.commit_device(coeffs.data_ptr(), rows * npr, enc, st, sync=False, borrow=True, into=c)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 10
        roots[plan] = c.get_root()
        print(json.dumps({"log_n": log_n, "rows": rows, "plan": plan, "ms_per_commit": round(dt * 1e3, 3),
                          "elems_per_s": rows * npr / dt}), flush=True)
        del c, enc
    assert roots["tile1024"] == roots["general"], "the two plans disagree"
    del coeffs
