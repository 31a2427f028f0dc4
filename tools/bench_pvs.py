#!/usr/bin/env python3
"""tools/bench_pvs.py [max_lgl] [kinds] -- the reference's own benchmark loops on the MI355X path, one JSON line per size,
for all three published Ligero rate series (doc/benchmark-results/20210807_64c_255bit_ligero_{hlf,dfl,isz}*.txt: rho = 1/2,
1/4 -- the timing test's default, lcpc-ligero-pc/src/tests.rs:59-69 --, 38/39) and Brakedown, up to 2^29 (tests.rs:83):
rough_bench (lcpc-ligero-pc/src/tests.rs:80-97, lcpc-brakedown-pc/src/tests.rs:171-190: mean commit time) and
prove_verify_size_bench (ligero tests.rs:102-170, brakedown tests.rs:98-167: mean prove time incl. bincode, mean verify
time, bincode proof bytes) for Ft255, len = 2^lgl, lgl = 13, 15, ... as there; encoder construction outside the timed
region as there.  The published 64-thread CPU numbers for the same loops are in BASELINE.md."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tools")]
import numpy as np
import torch

import bench_configs as B
import oracle_lib as O
from common import mk_transcript, powers
from lcpc_amd import LcCommit, LigeroEncoding, SdigEncoding, Transcript

N_ITERS = 10


RHO = {"ligero": (1, 2), "ligero_hlf": (1, 2), "ligero_dfl": (1, 4), "ligero_isz": (38, 39)}


def run(kind, lgl, fid=3):
    n = 1 << lgl
    L = fid + 1
    enc = LigeroEncoding.new(fid, n, rho=RHO[kind]) if kind in RHO else SdigEncoding.new(fid, n, 0)
    coeffs = B.rand_coeffs(n, L, lgl)
    st = torch.cuda.current_stream().cuda_stream
    c = LcCommit(enc)
    for _ in range(2):
        LcCommit.commit_device(coeffs.data_ptr(), n, enc, st, sync=True, into=c)
    def timed(fn):
        """mean (what the reference's loops report) and min over N_ITERS calls: right after a series that held tens of GB the
        first calls of the next one can stall for tens of ms while the freed memory is unmapped -- the min shows that"""
        ts = []
        for _ in range(N_ITERS):
            t0 = time.perf_counter()
            fn()
            ts.append(time.perf_counter() - t0)
        return sum(ts) / len(ts), min(ts)

    t_commit, t_commit_min = timed(lambda: LcCommit.commit_device(coeffs.data_ptr(), n, enc, st, sync=True, into=c))   # root on the host every time
    root = c.get_root()
    x = 0x1234567 + lgl
    inner = powers(O, fid, x, c.n_per_row)
    outer = powers(O, fid, x, c.n_rows, c.n_per_row)
    nco = enc.get_n_col_opens()
    pf = c.prove(outer, enc, mk_transcript(Transcript, root, nco))
    t_prove, t_prove_min = timed(lambda: c.prove(outer, enc, mk_transcript(Transcript, root, nco)))
    pf.verify(root, outer, inner, enc, mk_transcript(Transcript, root, nco))
    t_verify, t_verify_min = timed(lambda: pf.verify(root, outer, inner, enc, mk_transcript(Transcript, root, nco)))
    print(json.dumps({"enc": kind, "field": ("ft63", "ft127", "ft191", "ft255")[fid], "lgl": lgl, "dims": [c.n_rows, c.n_per_row, c.n_cols], "commit_ms": round(t_commit * 1e3, 3),
                      "prove_ms": round(t_prove * 1e3, 3), "verify_ms": round(t_verify * 1e3, 3),
                      "min_ms": [round(t_commit_min * 1e3, 3), round(t_prove_min * 1e3, 3), round(t_verify_min * 1e3, 3)], "proof_bytes": len(pf.to_bytes())}),
          flush=True)
    del coeffs, c, enc
    torch.cuda.empty_cache()


def main():
    max_lgl = int(sys.argv[1]) if len(sys.argv) > 1 else 29
    kinds = sys.argv[2].split(",") if len(sys.argv) > 2 else ("ligero_hlf", "ligero_dfl", "ligero_isz", "sdig")
    for kind in kinds:
        for lgl in range(13, max_lgl + 1, 2):
            if kind == "sdig" and lgl > 27:
                continue                       # (the reference's Brakedown series stops at 2^27 too)
            run(kind, lgl)


if __name__ == "__main__":
    main()
