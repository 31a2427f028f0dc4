#!/usr/bin/env python3
"""tools/fuzz_sharded.py [n_cases] -- random row-sharded commits (G shard contexts on one GPU, emulated all-gather, as in
tests/test_gpu_sharded.py): encoding (Ligero; every fifth case Brakedown, whose shards fall on both sides of the row-major /
position-major threshold), field, shape, row count and shard count drawn at random; every rank's root and full hashes array
must equal the unsharded oracle commitment.  One-off soak, not part of the test suite."""
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")]
import numpy as np
import torch

import oracle_lib as O
import test_gpu_sharded as T
from lcpc_amd import LcCommit, LigeroEncoding, SdigEncoding

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rnd = random.Random(4242)
for case in range(n_cases):
    fid = rnd.choice([0, 1, 2, 3, 3])
    L = O.limbs(fid)
    log_n = rnd.randrange(2, 16)          # up to 2^15 columns: the specialised two-pass kernels (K1s / K1n, canonical comm) under sharding
    n_cols = 1 << log_n
    n_per_row = rnd.randrange(1, n_cols)
    n_rows = rnd.randrange(1, max(2, min(700, (1 << 21) // n_cols)))
    G = rnd.choice([2, 3, 4, 5, 8])
    if case % 5 == 4:
        n_per_row = rnd.randrange(200, 3000)
        n_rows = rnd.randrange(1, 260)
        seed, code = rnd.randrange(1000), rnd.choice([1, 3, 3, 5])
        oenc = O.Encoding.sdig_from_dims(fid, n_per_row, 0, seed, code)
        _, _, n_cols = oenc.get_dims(n_per_row)
        coeffs = O.random_elems(fid, n_rows * n_per_row, rnd.randrange(1 << 30))
        dev = torch.from_numpy(coeffs.view(np.int64)).cuda().reshape(n_rows, n_per_row, L)
        roots, engines = T.run_sharded(lambda sh: SdigEncoding(fid, None, seed, code, 0, sh, _dims=(n_per_row, n_cols)), G, dev, n_rows)
        oc = O.Commit.commit(coeffs, oenc, n_threads=4)
        assert all(r == oc.get_root() for r in roots), (case, "sdig", fid, n_rows, n_per_row, G, seed, code)
        for eng in engines:
            assert (eng.cm.hashes() == oc.hashes()).all(), (case, "sdig hashes")
        continue
    coeffs = O.random_elems(fid, n_rows * n_per_row, rnd.randrange(1 << 30))
    dev = torch.from_numpy(coeffs.view(np.int64)).cuda().reshape(n_rows, n_per_row, L)
    roots, engines = T.run_sharded(lambda sh: LigeroEncoding.new_from_dims(fid, n_per_row, n_cols, shard=sh), G, dev, n_rows)
    oc = O.Commit.commit(coeffs, O.Encoding.ligero_from_dims(fid, n_per_row, n_cols), n_threads=4)
    assert all(r == oc.get_root() for r in roots), (case, fid, n_rows, n_per_row, n_cols, G)
    for eng in engines:
        assert (eng.cm.hashes() == oc.hashes()).all(), (case, "hashes")
    if case % 25 == 0:
        print("case", case, "ok", flush=True)
print("all", n_cases, "sharded cases ok")
