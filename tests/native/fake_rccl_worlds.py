#!/usr/bin/env python3
"""tests/native/fake_rccl_worlds.py -- the library's NATIVE exchange with N > 1 ranks on one GPU (run by tests/test_gpu_fake_rccl.py
with LCPC_RCCL_LIB pointing at tests/native/fake_rccl.cpp's library: ranks = host threads of this process).

Per case: G threads, each with its own sharded encoder (rank r of G), ONE communicator (lcpc_comm_init with a shared id), and
 1. lcpc_commit_sharded_device in sequence on the rank's stream -> every rank's root and whole `hashes` == the oracle's;
 2. the same with LCPC_COMMIT_ASYNC_TAIL on two commitments per rank, filled alternately with two polynomials and no host
    synchronisation, then a refill of each -> roots / hashes of the LAST polynomial committed into each;
 2b. lcpc_shard_exchange_probe (bench.py's wire probe) between commits: the commitment and the next commit are unaffected;
 3. lcpc_prove_sharded_rccl on every rank -> the oracle prover's bytes, identical on every rank;
for Ligero and Brakedown, the four fields (Ft191's shards cut at rows = 84 mod 128), node layouts with one, two and three nodes
per rank and ranks that own nothing.
Prints one line per case and "all ok"; any mismatch raises."""
import ctypes as C
import os
import sys
import threading

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")]
import numpy as np
import torch

import lcpc_amd
import oracle_lib as O
from common import mk_transcript
from lcpc_amd import LigeroEncoding, SdigEncoding, Transcript, _lib
from lcpc_amd.distributed import HipShardEngine

assert os.environ.get("LCPC_RCCL_LIB"), "run through tests/test_gpu_fake_rccl.py (LCPC_RCCL_LIB = the stand-in library)"

CASES = [
    # kind, fid, n_rows, n_per_row, n_cols, G
    ("ligero", 3, 512, 256, 512, 8),       # headline row count: 17 chunks, nodes 1,1,1,1,1,1,1,2 (one broadcast)
    ("ligero", 3, 512, 2048, 4096, 2),     # 2 ranks: 1 + 2 nodes; K1s rows
    ("ligero", 3, 1024, 128, 256, 8),      # C4's row count: 33 chunks
    ("ligero", 3, 1024, 2048, 4096, 4),     # K1s rows, 33 chunks over 4 ranks
    ("ligero", 3, 70, 64, 128, 4),         # 3 chunks over 4 ranks: one rank owns nothing
    ("ligero", 3, 20, 64, 128, 3),         # single chunk: two ranks own nothing
    ("ligero", 3, 40, 64, 128, 2),         # 2 chunks
    ("ligero", 0, 3000, 128, 256, 3),      # ft63: 24 chunks over 3 ranks (8 each: one node per rank)
    ("ligero", 0, 2900, 2048, 4096, 5),     # ft63 on K1n, 23 chunks over 5 ranks: up to three nodes per rank
    ("ligero", 1, 700, 64, 128, 4),        # ft127
    ("ligero", 2, 700, 64, 128, 3),        # ft191: cuts at chunks 5 and 11 (rows 212 and 468)
    ("ligero", 2, 1500, 2048, 4096, 8),    # ft191 on K1n: 36 chunks over 8 ranks
    ("sdig", 3, 140, 300, 0, 4),           # Brakedown: position-major shards (>= 24 local rows)
    ("sdig", 3, 70, 3000, 0, 2),            # Brakedown, 4500-odd columns
]
if len(sys.argv) > 1:
    CASES = [CASES[int(a)] for a in sys.argv[1:]]

lib = _lib.lib()


def run_case(kind, fid, n_rows, n_per_row, n_cols, G):
    L = O.limbs(fid)
    if kind == "ligero":
        oenc = O.Encoding.ligero_from_dims(fid, n_per_row, n_cols)
        mk = lambda sh: LigeroEncoding.new_from_dims(fid, n_per_row, n_cols, shard=sh)
    else:
        oenc = O.Encoding.sdig_from_dims(fid, n_per_row, 0, 11, 3)
        _, _, n_cols = oenc.get_dims(n_per_row)
        mk = lambda sh: SdigEncoding(fid, None, 11, 3, 0, sh, _dims=(n_per_row, n_cols))
    polys = [O.random_elems(fid, n_rows * n_per_row, 800 + i) for i in range(2)]
    ocs = [O.Commit.commit(p, oenc, n_threads=8) for p in polys]
    devs = [torch.from_numpy(p.view(np.int64)).cuda().reshape(n_rows, n_per_row, L) for p in polys]
    outer = O.random_elems(fid, n_rows, 810)
    idb = (C.c_uint8 * 128)()
    assert lib.lcpc_comm_unique_id(idb) == 0
    assert bytes(idb)[8:] == b"\0" * 120 and int.from_bytes(bytes(idb)[:8], "little") > 0, "not the stand-in library: ids of the real RCCL are opaque"
    encs = [mk((g, G)) for g in range(G)]
    n_open = encs[0].get_n_col_opens()
    opfs = [oc.prove(outer, oenc, mk_transcript(O.Transcript, oc.get_root(), n_open))[0] for oc in ocs]
    errs, bar = [], threading.Barrier(G)

    def rank(g):
        try:
            torch.cuda.set_device(0)
            enc = encs[g]
            a, b = HipShardEngine(enc), HipShardEngine(enc)
            enc._check(lib.lcpc_comm_init(enc._h, idb, g, G))
            rb, re, _, _, _ = a.layout(n_rows)
            loc = [d[rb:re].contiguous() for d in devs]
            st = torch.cuda.Stream()
            st.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(st):
                # 1. in sequence on the rank's stream, with and without borrowed coefficients
                for borrow in (False, True):
                    r = a.commit_native(loc[0], n_rows, want_root=True, borrow=borrow)
                    assert r == ocs[0].get_root(), "rank %d: sequential root" % g
                assert (a.cm.hashes() == ocs[0].hashes()).all(), "rank %d: sequential hashes" % g
                # 2. async tails on two commitments, alternately, no host synchronisation in between
                for rep in range(2):
                    a.commit_native(loc[0], n_rows, want_root=False, async_tail=True)
                    b.commit_native(loc[1], n_rows, want_root=False, async_tail=True)
                    a.commit_native(loc[1], n_rows, want_root=False, async_tail=True)      # refill: behind a's own tail
                    b.commit_native(loc[0], n_rows, want_root=False, async_tail=True)
                assert a.cm.get_root() == ocs[1].get_root() and b.cm.get_root() == ocs[0].get_root(), "rank %d: async roots" % g
            st.synchronize()
            a.cm._refresh(); b.cm._refresh()
            assert (a.cm.hashes() == ocs[1].hashes()).all() and (b.cm.hashes() == ocs[0].hashes()).all(), "rank %d: async hashes" % g
            # 2b. the measurement hook: the exchange of a's last commit alone, twice more (collective: every rank, same order).  What
            #     arrives lands in the gathered area only: the finished commitment is untouched, and so is the next commit
            n_in = a.exchange_probe()
            assert n_in == a.exchange_probe()
            torch.cuda.current_stream().synchronize()
            assert n_in % (n_cols * 32) == 0 and (G - 1) * n_cols * 32 <= n_in + n_cols * 32 * 2, "rank %d: probe bytes %d" % (g, n_in)
            assert (a.cm.hashes() == ocs[1].hashes()).all(), "rank %d: hashes after the probe" % g
            assert a.commit_native(loc[1], n_rows, want_root=True) == ocs[1].get_root(), "rank %d: commit after the probe" % g
            # 3. sharded prove on both commitments (three all-gathers each; same order on every rank)
            for eng, k in ((a, 1), (b, 0)):
                data, _ = eng.prove_native(outer, mk_transcript(Transcript, ocs[k].get_root(), n_open))
                assert data == opfs[k], "rank %d: proof bytes" % g
        except BaseException as e:          # never leave the other ranks waiting inside a collective without a trace
            errs.append((g, repr(e)))
            bar.abort()

    th = [threading.Thread(target=rank, args=(g,), daemon=True) for g in range(G)]
    for t in th:
        t.start()
    for t in th:
        t.join(180)
    alive = [t for t in th if t.is_alive()]
    if alive or errs:
        print("FAILED case %r: %s%s" % ((kind, fid, n_rows, n_per_row, n_cols, G), errs, " (ranks hung)" if alive else ""), flush=True)
        os._exit(1)                         # (ranks blocked inside a collective cannot be joined)
    print("ok", kind, fid, n_rows, n_per_row, n_cols, "G=%d" % G, flush=True)


for case in CASES:
    run_case(*case)
print("all ok")
