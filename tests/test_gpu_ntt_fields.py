"""K1n (ntt_lns.hip): the shape-specialised lazy-limb row NTT of Ft63 / Ft127 / Ft191 -- two-pass plans on 1024-element
tiles, n_cols = 2^11 .. 2^20 wherever the general plan needs more than one pass.  For each: commit (coeffs copy fused into
pass 1, ragged last row) and encode_rows against the oracle at the rates the reference uses (1/2, 1/4, 38/39) and 3/4; the
general kernel (LCPC_NTT_GENERAL=1) must give the same bytes; inputs that push the signed lazy-limb bounds (all p-1,
saturated limbs, alternating 0 / p-1)."""
import os

import numpy as np
import pytest

from lcpc_amd import LcCommit, LigeroEncoding

pytestmark = pytest.mark.gpu

FIRST_TWO_PASS = {0: 13, 1: 12, 2: 11}      # smallest log2 n_cols whose general plan has two passes (ctx.cpp plan_passes)


def _shapes():
    out = []
    for fid in (0, 1, 2):
        for log_n in range(FIRST_TWO_PASS[fid], 19):
            out.append((fid, log_n))
    return out


@pytest.mark.parametrize("fid,log_n", _shapes())
@pytest.mark.parametrize("rate", ["1/2", "1/4", "38/39", "3/4", "1/2-"])
def test_commit_all_two_pass_shapes_small_fields(oracle, fid, log_n, rate):
    O = oracle
    L = fid + 1
    n_cols = 1 << log_n
    n_per_row, rho = {"1/2": (n_cols // 2, (1, 2)), "1/4": (n_cols // 4, (1, 4)), "38/39": (n_cols * 38 // 39, (38, 39)),
                      "3/4": (n_cols * 3 // 4, (3, 4)), "1/2-": (n_cols // 2 - 3, (1, 2))}[rate]
    n = 2 * n_per_row + max(1, n_per_row // 3)               # 3 rows, ragged
    coeffs = O.random_elems(fid, n, log_n * 7 + len(rate) + fid)
    enc = LigeroEncoding.new_from_dims(fid, n_per_row, n_cols, rho=rho)
    oenc = O.Encoding.ligero_from_dims(fid, n_per_row, n_cols, rho=rho)
    c = LcCommit.commit(coeffs, enc)
    oc = O.Commit.commit(coeffs, oenc, n_threads=4)
    assert c.get_root() == oc.get_root()
    assert (c.hashes() == oc.hashes()).all()
    assert (c.comm() == oc.comm()).all()
    assert (c.coeffs() == oc.coeffs()).all()
    rows = np.zeros((2 * n_cols, L), np.uint64)
    rows[:n_per_row] = coeffs[n_per_row:2 * n_per_row]
    rows[n_cols:n_cols + (n - 2 * n_per_row)] = coeffs[2 * n_per_row:]
    assert (enc.encode(rows) == oc.comm()[n_cols:]).all()


@pytest.mark.parametrize("fid,log_n", [(0, 19), (0, 20), (1, 19), (1, 20), (2, 19), (2, 20)])
@pytest.mark.parametrize("rate", ["1/2", "38/39"])
def test_commit_long_rows_small_fields(oracle, fid, log_n, rate):
    """2^19 / 2^20 columns: first passes of 9 / 10 stages whose runs are 2 / 1 elements (8 bytes for Ft63 at 2^20): the tiles
    that share cache lines run back to back on one XCD (ctx.cpp ntt_tile_group), which keeps these ahead of the general
    kernel's three-pass plan"""
    O = oracle
    n_cols = 1 << log_n
    n_per_row, rho = {"1/2": (n_cols // 2, (1, 2)), "38/39": (n_cols * 38 // 39, (38, 39))}[rate]
    n = n_per_row + n_per_row // 5
    coeffs = O.random_elems(fid, n, log_n + fid)
    enc = LigeroEncoding.new_from_dims(fid, n_per_row, n_cols, rho=rho)
    oenc = O.Encoding.ligero_from_dims(fid, n_per_row, n_cols, rho=rho)
    c = LcCommit.commit(coeffs, enc)
    oc = O.Commit.commit(coeffs, oenc, n_threads=4)
    assert c.get_root() == oc.get_root() and (c.hashes() == oc.hashes()).all()
    assert (c.comm() == oc.comm()).all() and (c.coeffs() == oc.coeffs()).all()


@pytest.mark.parametrize("fid,log_n", [(0, 13), (0, 16), (0, 18), (1, 12), (1, 16), (2, 11), (2, 17)])
def test_general_kernel_agrees_small_fields(oracle, fid, log_n):
    O = oracle
    n_cols, n_per_row = 1 << log_n, 1 << (log_n - 1)
    coeffs = O.random_elems(fid, 5 * n_per_row - 9, 3 + log_n + fid)
    c = LcCommit.commit(coeffs, LigeroEncoding.new_from_dims(fid, n_per_row, n_cols))
    os.environ["LCPC_NTT_GENERAL"] = "1"
    try:
        enc_g = LigeroEncoding.new_from_dims(fid, n_per_row, n_cols)
    finally:
        del os.environ["LCPC_NTT_GENERAL"]
    g = LcCommit.commit(coeffs, enc_g)
    assert (g.comm() == c.comm()).all() and (g.hashes() == c.hashes()).all()
    assert g.get_root() == O.Commit.commit(coeffs, O.Encoding.ligero_from_dims(fid, n_per_row, n_cols), n_threads=4).get_root()


@pytest.mark.parametrize("fid", [0, 1, 2])
def test_lazy_limb_range_stress_small_fields(oracle, fid):
    import pyref as P
    O = oracle
    F = {0: P.FT63, 1: P.FT127, 2: P.FT191}[fid]
    L = fid + 1
    W, N = {0: (26, 3), 1: (29, 5), 2: (29, 7)}[fid]
    sat = sum(((1 << W) - 1) << (W * k) for k in range(N)) % F.p
    pats = [[F.p - 1], [(F.p - 1) // 2], [0, F.p - 1], [sat, F.p - 2, 1], [F.p - 1, F.p - 1, F.p - 1, 0]]
    k0 = FIRST_TWO_PASS[fid]
    for log_n, n_per_row in ((k0, 1 << (k0 - 1)), (k0 + 1, (1 << (k0 + 1)) * 38 // 39), (16, 1 << 15), (17, 1 << 16)):
        n = 1 << log_n
        enc = LigeroEncoding.new_from_dims(fid, n_per_row, n, rho=(38, 39) if n_per_row > n // 2 else (1, 2))
        oenc = O.Encoding.ligero_from_dims(fid, n_per_row, n, rho=(38, 39) if n_per_row > n // 2 else (1, 2))
        rows = np.zeros((len(pats), n, L), np.uint64)
        for r, pat in enumerate(pats):
            m = O.to_mont(fid, pat)
            reps = (n_per_row + len(pat) - 1) // len(pat)
            rows[r, :n_per_row] = np.tile(m, (reps, 1))[:n_per_row]
        got = enc.encode(rows).reshape(len(pats), n, L)
        for r in range(len(pats)):
            assert (got[r] == oenc.encode(rows[r].copy())).all(), (log_n, n_per_row, r)


@pytest.mark.parametrize("fid,log_n", [(0, 13), (0, 14), (0, 17), (0, 20), (1, 13), (1, 16), (1, 19), (2, 14), (2, 18), (2, 20), (3, 13), (3, 14), (3, 15),
                                       (3, 17), (3, 18), (3, 19), (3, 20)])
def test_tile_group_order_row_counts(oracle, fid, log_n):
    """the first pass's workgroup -> tile mapping (2^g neighbouring tiles back to back on one XCD, g by the run length; the row
    count enters the mapping) is a permutation of the work: 1, 3 and 5 rows give the oracle's commitment and keep the coeffs copy"""
    O = oracle
    L = fid + 1
    n_cols, n_per_row = 1 << log_n, 1 << (log_n - 1)
    enc = LigeroEncoding.new_from_dims(fid, n_per_row, n_cols)
    oenc = O.Encoding.ligero_from_dims(fid, n_per_row, n_cols)
    for n_rows in (1, 3, 5) if log_n <= 18 else (3,):
        coeffs = O.random_elems(fid, n_rows * n_per_row - 7, 5 * log_n + fid + n_rows)
        c = LcCommit.commit(coeffs, enc)
        oc = O.Commit.commit(coeffs, oenc, n_threads=4)
        assert c.get_root() == oc.get_root() and (c.comm() == oc.comm()).all() and (c.hashes() == oc.hashes()).all()
        assert (c.coeffs() == coeffs_padded(coeffs, n_rows, n_per_row, L)).all()


def coeffs_padded(coeffs, n_rows, n_per_row, L):
    out = np.zeros((n_rows * n_per_row, L), np.uint64)
    out[:len(coeffs)] = coeffs
    return out


@pytest.mark.parametrize("fid,log_n,rate", [(0, 21, "1/2"), (0, 22, "38/39"), (0, 23, "1/2"), (1, 21, "1/2"), (1, 21, "38/39"), (1, 22, "1/4"), (2, 21, "1/2-"),
                                            (2, 22, "1/2"), (2, 21, "3/4")])
def test_commit_three_pass_shapes_small_fields(oracle, fid, log_n, rate):
    """2^21 .. 2^26 columns on K1n: the three-pass plan of tests/test_gpu_ntt_shapes.py::test_commit_three_pass_shapes (first-pass
    kernel over the whole rows, then the 2^20-point two-pass plan per block, canonical output in block 0 only) for Ft63 / Ft127 /
    Ft191: commit (2 rows, the second ragged) and encode_rows against the oracle, and the general kernel's bytes."""
    O = oracle
    L = fid + 1
    n_cols = 1 << log_n
    n_per_row, rho = {"1/2": (n_cols // 2, (1, 2)), "1/4": (n_cols // 4, (1, 4)), "38/39": (n_cols * 38 // 39, (38, 39)),
                      "3/4": (n_cols * 3 // 4, (3, 4)), "1/2-": (n_cols // 2 - 3, (1, 2))}[rate]
    n = n_per_row + max(1, n_per_row // 3)
    coeffs = O.random_elems(fid, n, log_n * 3 + len(rate) + fid)
    enc = LigeroEncoding.new_from_dims(fid, n_per_row, n_cols, rho=rho)
    oenc = O.Encoding.ligero_from_dims(fid, n_per_row, n_cols, rho=rho)
    c = LcCommit.commit(coeffs, enc)
    oc = O.Commit.commit(coeffs, oenc, n_threads=8)
    assert c.get_root() == oc.get_root() and (c.hashes() == oc.hashes()).all()
    assert (c.comm() == oc.comm()).all() and (c.coeffs() == oc.coeffs()).all()
    rows = np.zeros((n_cols, L), np.uint64)
    rows[:n - n_per_row] = coeffs[n_per_row:]
    assert (enc.encode(rows) == oc.comm()[n_cols:]).all()
    os.environ["LCPC_NTT_GENERAL"] = "1"
    try:
        enc_g = LigeroEncoding.new_from_dims(fid, n_per_row, n_cols, rho=rho)
    finally:
        del os.environ["LCPC_NTT_GENERAL"]
    g = LcCommit.commit(coeffs, enc_g)
    assert (g.comm() == c.comm()).all() and (g.hashes() == c.hashes()).all()
