#!/usr/bin/env python3
"""tools/bench_configs.py -- timings of the other BASELINE.json configs (C1..C5) on one MI355X, one JSON line each.
Not the driver's bench (that is bench.py): this is the evidence quoted in DESIGN.md section 6."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import torch

import lcpc_amd
from lcpc_amd import LcCommit, LigeroEncoding, SdigEncoding, Transcript


BORROW = os.environ.get("LCPC_BENCH_COPY") is None      # LcCommit.coeffs aliases the input unless LCPC_BENCH_COPY is set


def rand_coeffs(n, L, seed):
    g = torch.Generator(device="cuda")
    g.manual_seed(seed)
    t = torch.randint(-(1 << 63), (1 << 63) - 1, (n, L), dtype=torch.int64, device="cuda", generator=g)
    t[:, L - 1] &= (1 << 62) - 1
    return t


def time_commit(name, enc, n, L, iters=5):
    coeffs = rand_coeffs(n, L, 1)
    st = torch.cuda.current_stream().cuda_stream
    c = LcCommit(enc)                       # one LcCommit object, refilled (no allocation inside the loop)
    # >= 0.1 s of warm-up, then >= 0.2 s (and >= iters) back-to-back commits: after the idle set-up the device needs tens of
    # milliseconds to come back to working clocks (bench.py's setup steps; tools/bench_fields.py)
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.1:
        for _ in range(2):
            LcCommit.commit_device(coeffs.data_ptr(), n, enc, st, sync=False, borrow=BORROW, into=c)
        torch.cuda.synchronize()
    reps = 0
    t0 = time.perf_counter()
    while reps < iters or time.perf_counter() - t0 < 0.2:
        for _ in range(iters):
            LcCommit.commit_device(coeffs.data_ptr(), n, enc, st, sync=False, borrow=BORROW, into=c)
        torch.cuda.synchronize()
        reps += iters
    dt = (time.perf_counter() - t0) / reps
    c.set_timing(True)
    LcCommit.commit_device(coeffs.data_ptr(), n, enc, st, sync=True, borrow=BORROW, into=c)
    tm = c.timings()
    c.set_timing(False)
    print(json.dumps({"config": name, "n_coeffs": n, "dims": [c.n_rows, c.n_per_row, c.n_cols], "ms_per_commit": round(dt * 1e3, 3),
                      "elems_per_s": n / dt, "group_ms": {"encode": round(tm.encode_ms, 3), "hash": round(tm.hash_ms, 3),
                                                          "merkle": round(tm.merkle_ms, 3)}}), flush=True)
    return c, coeffs


def main():
    which = sys.argv[1:] or ["c1", "c2", "c3", "head", "c5", "c5s", "c4"]
    if "c1" in which:
        time_commit("C1 ligero ft63 2^16", LigeroEncoding.new(0, 1 << 16), 1 << 16, 1, 20)
    if "c2" in which:
        time_commit("C2 ligero ft255 2^24", LigeroEncoding.new(3, 1 << 24), 1 << 24, 4)
    if "c3" in which:
        t0 = time.perf_counter()
        enc = SdigEncoding.new(3, 1 << 24, 0)
        print(json.dumps({"config": "C3 matgen+upload (SdigEncoding::new)", "seconds": round(time.perf_counter() - t0, 3)}), flush=True)
        time_commit("C3 brakedown ft255 2^24", enc, 1 << 24, 4)
        del enc
    if "head" in which or "c5" in which:
        enc = LigeroEncoding.new(3, 1 << 26)
        c, coeffs = time_commit("headline ligero ft255 2^26", enc, 1 << 26, 4)
        if "c5" in which:
            import oracle_lib as O
            from common import mk_transcript, powers
            import pyref as P
            x = 0x123456789abcdef % P.FT255.p
            inner = powers(O, 3, x, c.n_per_row)
            outer = powers(O, 3, x, c.n_rows, c.n_per_row)
            root = c.get_root()
            st = torch.cuda.current_stream().cuda_stream
            tp, tv = [], []
            for rep in range(4):                                 # rep 0 = first use (allocations, helper threads): not counted
                # prove follows commit in the reference's flow (tests.rs:243-262): re-commit right before it so the
                # collapse kernel does not start on a GPU that dropped its clocks while the host prepared the tensors
                # (a 0.6 ms kernel takes 8-20 ms on an idle-clocked device)
                LcCommit.commit_device(coeffs.data_ptr(), 1 << 26, enc, st, sync=True, into=c)
                t0 = time.perf_counter()
                pf = c.prove(outer, enc, mk_transcript(Transcript, root, enc.get_n_col_opens()))
                t1 = time.perf_counter()
                ev = pf.verify(root, outer, inner, enc, mk_transcript(Transcript, root, enc.get_n_col_opens()))
                t2 = time.perf_counter()
                if rep:
                    tp.append(t1 - t0); tv.append(t2 - t1)
            t_prove, t_verify = sum(tp) / len(tp), sum(tv) / len(tv)
            # GPU part of prove only: fused collapse of 2 tensors + open 309 columns
            tens = np.stack([outer, outer])
            c.eval_outer(tens)                                   # first use: output allocation
            c.open_columns(pf.cols_opened)
            LcCommit.commit_device(coeffs.data_ptr(), 1 << 26, enc, st, sync=True, into=c)     # GPU at working clocks (see above)
            t0 = time.perf_counter()
            c.eval_outer(tens)
            t_col = time.perf_counter() - t0
            t0 = time.perf_counter()
            c.open_columns(pf.cols_opened)
            t_open = time.perf_counter() - t0
            print(json.dumps({"config": "C5 prove+verify ft255 2^26", "prove_ms": round(t_prove * 1e3, 2), "verify_ms": round(t_verify * 1e3, 2),
                              "prove_min_ms": round(min(tp) * 1e3, 2), "verify_min_ms": round(min(tv) * 1e3, 2), "iters": len(tp),
                              "collapse2_incl_copies_ms": round(t_col * 1e3, 2), "open309_incl_copies_ms": round(t_open * 1e3, 2),
                              "proof_bytes": len(pf.to_bytes())}), flush=True)
        del enc, c, coeffs
    if "c5s" in which:
        # the SHARDED prover's code path at world = 1 (RCCL communicator of one rank: the three all-gathers are self-copies):
        # what the exchange plumbing of lcpc_prove_sharded_rccl costs next to the plain prover on the same commitment
        import oracle_lib as O
        from common import mk_transcript, powers
        import pyref as P
        from lcpc_amd.distributed import HipShardEngine
        n = 1 << 26
        nr, npr, nc = lcpc_amd.static_get_dims(3, lcpc_amd.ENC_LIGERO, n)
        enc = LigeroEncoding.new_from_dims(3, npr, nc, shard=(0, 1))
        eng = HipShardEngine(enc)
        eng.comm_init()
        coeffs = rand_coeffs(n, 4, 1)
        x = 0x123456789abcdef % P.FT255.p
        outer = powers(O, 3, x, nr, npr)
        ts, tp = [], []
        for rep in range(4):
            root = eng.commit_native(coeffs, nr)
            t0 = time.perf_counter()
            data, _ = eng.prove_native(outer, mk_transcript(Transcript, root, enc.get_n_col_opens()))
            t1 = time.perf_counter()
            pf = eng.cm.prove(outer, enc, mk_transcript(Transcript, root, enc.get_n_col_opens()))
            t2 = time.perf_counter()
            assert pf.to_bytes() == data
            if rep:
                ts.append(t1 - t0); tp.append(t2 - t1)
        print(json.dumps({"config": "C5 prove through the sharded path, world = 1 (RCCL self-exchange), ft255 2^26",
                          "sharded_prove_ms": round(sum(ts) / len(ts) * 1e3, 2), "sharded_prove_min_ms": round(min(ts) * 1e3, 2),
                          "plain_prove_same_commit_ms": round(sum(tp) / len(tp) * 1e3, 2), "proof_bytes_equal": True}), flush=True)
        del eng, enc, coeffs
    if "c4" in which:
        torch.cuda.empty_cache()
        time_commit("C4-on-1-GPU ligero ft255 2^28", LigeroEncoding.new(3, 1 << 28), 1 << 28, 4, 3)


if __name__ == "__main__":
    main()
