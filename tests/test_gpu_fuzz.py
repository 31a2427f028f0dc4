"""Seeded random sweep of commit shapes against the oracle: field x rate x length for Ligero (every pass plan the
planner can produce up to 2^18 columns, all four fields (the specialised two-pass kernels K1s / K1n and the general kernel), ragged last rows, 1..600 rows), field x code x length for Brakedown.  Each
case checks comm, coeffs, every digest of the tree and one collapse; a few also run prove and compare proof bytes."""
import random

import numpy as np
import pytest

from common import mk_transcript
from lcpc_amd import LcCommit, LigeroEncoding, SdigEncoding, Transcript

pytestmark = pytest.mark.gpu


def _check(O, c, oc, fid, enc, oenc, rnd, do_prove):
    assert (c.n_rows, c.n_per_row, c.n_cols) == (oc.n_rows, oc.n_per_row, oc.n_cols)
    assert (c.comm() == oc.comm()).all()
    assert (c.coeffs() == oc.coeffs()).all()
    assert (c.hashes() == oc.hashes()).all()
    t = O.random_elems(fid, c.n_rows, rnd.randrange(1 << 30))
    assert (c.eval_outer(t) == oc.collapse(t)).all()
    if do_prove:
        root = c.get_root()
        pf = c.prove(t, enc, mk_transcript(Transcript, root, enc.get_n_col_opens()))
        opf, _ = oc.prove(t, oenc, mk_transcript(O.Transcript, root, enc.get_n_col_opens()))
        assert pf.to_bytes() == opf


@pytest.mark.parametrize("seed", range(8))
def test_fuzz_ligero(oracle, seed):
    O = oracle
    rnd = random.Random(1000 + seed)
    for i in range(20):
        fid = rnd.choice([0, 1, 2, 3, 3, 3])
        rho = rnd.choice([(1, 2), (1, 2), (1, 4), (3, 4), (38, 39)])
        log_n = rnd.randrange(1, 19)
        n_cols = 1 << log_n
        n_per_row = max(1, min(n_cols - 1, n_cols * rho[0] // rho[1] - rnd.choice([0, 0, 1, 3])))
        max_rows = max(1, min(600, (1 << 19) // n_cols))
        n_rows = rnd.randrange(1, max_rows + 1)
        n = n_rows * n_per_row - rnd.randrange(0, n_per_row)
        enc = LigeroEncoding.new_from_dims(fid, n_per_row, n_cols, rho)
        oenc = O.Encoding.ligero_from_dims(fid, n_per_row, n_cols, rho)
        coeffs = O.random_elems(fid, n, rnd.randrange(1 << 30))
        c = LcCommit.commit(coeffs, enc)
        oc = O.Commit.commit(coeffs, oenc, n_threads=4)
        _check(O, c, oc, fid, enc, oenc, rnd, do_prove=(i % 4 == 0))


@pytest.mark.parametrize("seed", range(4))
def test_fuzz_brakedown(oracle, seed):
    O = oracle
    rnd = random.Random(2000 + seed)
    for i in range(6):
        fid = rnd.choice([0, 1, 2, 3, 3])
        code = rnd.randrange(1, 7)
        n_per_row = rnd.randrange(60, 3000)
        n_rows = rnd.choice([1, 2, 7, 15, 16, 17, 23, 24, 25, 40, 64, 65, 90, 130])
        n = n_rows * n_per_row - rnd.randrange(0, n_per_row)
        mseed = rnd.randrange(1 << 40)
        oenc = O.Encoding.sdig_from_dims(fid, n_per_row, 0, mseed, code)
        _, _, n_cols = oenc.get_dims(n_per_row)
        enc = SdigEncoding.new_from_dims(fid, n_per_row, n_cols, mseed, code)
        coeffs = O.random_elems(fid, n, rnd.randrange(1 << 30))
        c = LcCommit.commit(coeffs, enc)
        oc = O.Commit.commit(coeffs, oenc, n_threads=4)
        _check(O, c, oc, fid, enc, oenc, rnd, do_prove=(i == 0))
