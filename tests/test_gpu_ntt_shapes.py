"""Every shape of the specialised Ft255 row NTT (ntt_l9s.hip): two-pass plans, n_cols = 2^11 .. 2^20, i.e. first passes of
1 .. 10 stages (odd counts peel a radix-2 round at stage 0) on 1 .. 512-element runs, last pass of 10 stages.  For each:
commit (canonical-output path, coeffs copy fused into pass 1, ragged last row) and encode_rows (Montgomery path) against
the oracle, at the rates the reference uses (1/2 default, 1/4 timing test, 38/39 and 3/4: no zero half / partly zero).
The general kernel (LCPC_NTT_GENERAL=1) must give the same bytes.  Three-pass plans (2^21 .. 2^24 columns) at the end."""
import os

import numpy as np
import pytest

from lcpc_amd import LcCommit, LigeroEncoding

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("log_n", list(range(11, 21)))
@pytest.mark.parametrize("rate", ["1/2", "1/4", "38/39", "3/4", "1/2-"])
def test_commit_all_two_pass_shapes(oracle, log_n, rate):
    O, fid = oracle, 3
    n_cols = 1 << log_n
    n_per_row, rho = {"1/2": (n_cols // 2, (1, 2)), "1/4": (n_cols // 4, (1, 4)), "38/39": (n_cols * 38 // 39, (38, 39)),
                      "3/4": (n_cols * 3 // 4, (3, 4)), "1/2-": (n_cols // 2 - 3, (1, 2))}[rate]
    n = 2 * n_per_row + max(1, n_per_row // 3)               # 3 rows, ragged
    coeffs = O.random_elems(fid, n, log_n * 7 + len(rate))
    enc = LigeroEncoding.new_from_dims(fid, n_per_row, n_cols, rho=rho)
    oenc = O.Encoding.ligero_from_dims(fid, n_per_row, n_cols, rho=rho)
    c = LcCommit.commit(coeffs, enc)
    oc = O.Commit.commit(coeffs, oenc, n_threads=4)
    assert c.get_root() == oc.get_root()
    assert (c.hashes() == oc.hashes()).all()
    assert (c.comm() == oc.comm()).all()
    assert (c.coeffs() == oc.coeffs()).all()
    # LcEncoding::encode on its own (no canonical-output conversion), rows 1 and 2
    rows = np.zeros((2 * n_cols, 4), np.uint64)
    rows[:n_per_row] = coeffs[n_per_row:2 * n_per_row]
    rows[n_cols:n_cols + (n - 2 * n_per_row)] = coeffs[2 * n_per_row:]
    assert (enc.encode(rows) == oc.comm()[n_cols:]).all()


@pytest.mark.parametrize("log_n", [11, 14, 17, 18, 19, 20])
def test_general_kernel_agrees(oracle, log_n):
    O, fid = oracle, 3
    n_cols, n_per_row = 1 << log_n, 1 << (log_n - 1)
    coeffs = O.random_elems(fid, 5 * n_per_row - 9, 3 + log_n)
    c = LcCommit.commit(coeffs, LigeroEncoding.new_from_dims(fid, n_per_row, n_cols))
    os.environ["LCPC_NTT_GENERAL"] = "1"
    try:
        enc_g = LigeroEncoding.new_from_dims(fid, n_per_row, n_cols)
    finally:
        del os.environ["LCPC_NTT_GENERAL"]
    g = LcCommit.commit(coeffs, enc_g)
    assert (g.comm() == c.comm()).all() and (g.hashes() == c.hashes()).all()
    assert g.get_root() == O.Commit.commit(coeffs, O.Encoding.ligero_from_dims(fid, n_per_row, n_cols), n_threads=4).get_root()


@pytest.mark.parametrize("log_n", [11, 12, 15, 18, 19])
@pytest.mark.parametrize("mid_mb", ["0", "1", "64"])
def test_limb_intermediate_modes_agree(oracle, log_n, mid_mb):
    """the 36-byte 29-bit-limb intermediate between the two passes (default: whole commitment in one batch) against the
    packed intermediate in comm itself (LCPC_NTT_MID_MAX_MB=0; the default above 2^15 columns), forced on (64 MiB) and in row batches (1 MiB of intermediate: 7 rows at
    2^12 columns, one row at a time or none at all above 2^14 -- then the packed path must take over)"""
    O, fid = oracle, 3
    n_cols, n_per_row = 1 << log_n, 1 << (log_n - 1)
    n_rows = 9 if log_n <= 15 else 3
    coeffs = O.random_elems(fid, n_rows * n_per_row - 5, 11 + log_n)
    enc = LigeroEncoding.new_from_dims(fid, n_per_row, n_cols)
    c = LcCommit.commit(coeffs, enc)
    os.environ["LCPC_NTT_MID_MAX_MB"] = mid_mb           # (switches are read once, when an encoder is created)
    try:
        enc_m = LigeroEncoding.new_from_dims(fid, n_per_row, n_cols)
    finally:
        del os.environ["LCPC_NTT_MID_MAX_MB"]
    g = LcCommit.commit(coeffs, enc_m)
    rows = np.zeros((n_rows * n_cols, 4), np.uint64)
    for r in range(n_rows):
        seg = coeffs[r * n_per_row:(r + 1) * n_per_row]
        rows[r * n_cols:r * n_cols + len(seg)] = seg
    e = enc_m.encode(rows)
    oc = O.Commit.commit(coeffs, O.Encoding.ligero_from_dims(fid, n_per_row, n_cols), n_threads=4)
    assert (g.comm() == c.comm()).all() and (g.hashes() == c.hashes()).all() and (g.coeffs() == c.coeffs()).all()
    assert g.get_root() == oc.get_root() and (c.comm() == oc.comm()).all()
    assert (e == oc.comm()).all()


@pytest.mark.parametrize("log_n,rate", [(21, "1/2"), (21, "38/39"), (21, "1/4"), (21, "1/2-"), (22, "1/2"), (22, "38/39"), (23, "1/2"), (23, "3/4"), (24, "1/2")])
def test_commit_three_pass_shapes(oracle, log_n, rate):
    """2^21 .. 2^26 columns (reachable on a 288 GB device: the reference's 38/39-rate dims at 2^30 .. 2^32 coefficients): s0 =
    log_n - 20 stages with the first-pass kernel over the whole rows, then the 2^20-point two-pass plan on each of the 2^s0 blocks
    of a row, canonical output only in block 0.  Commit (2 rows, the second ragged) and encode_rows against the oracle; the
    general kernel's three-pass plan (LCPC_NTT_GENERAL=1) must give the same bytes."""
    O, fid = oracle, 3
    n_cols = 1 << log_n
    n_per_row, rho = {"1/2": (n_cols // 2, (1, 2)), "1/4": (n_cols // 4, (1, 4)), "38/39": (n_cols * 38 // 39, (38, 39)),
                      "3/4": (n_cols * 3 // 4, (3, 4)), "1/2-": (n_cols // 2 - 3, (1, 2))}[rate]
    n = n_per_row + max(1, n_per_row // 3)
    coeffs = O.random_elems(fid, n, log_n * 5 + len(rate))
    enc = LigeroEncoding.new_from_dims(fid, n_per_row, n_cols, rho=rho)
    oenc = O.Encoding.ligero_from_dims(fid, n_per_row, n_cols, rho=rho)
    c = LcCommit.commit(coeffs, enc)
    oc = O.Commit.commit(coeffs, oenc, n_threads=8)
    assert c.get_root() == oc.get_root()
    assert (c.hashes() == oc.hashes()).all()
    assert (c.comm() == oc.comm()).all()
    assert (c.coeffs() == oc.coeffs()).all()
    rows = np.zeros((n_cols, 4), np.uint64)
    rows[:n - n_per_row] = coeffs[n_per_row:]
    assert (enc.encode(rows) == oc.comm()[n_cols:]).all()
    if log_n <= 22:
        os.environ["LCPC_NTT_GENERAL"] = "1"
        try:
            enc_g = LigeroEncoding.new_from_dims(fid, n_per_row, n_cols, rho=rho)
        finally:
            del os.environ["LCPC_NTT_GENERAL"]
        g = LcCommit.commit(coeffs, enc_g)
        assert (g.comm() == c.comm()).all() and (g.hashes() == c.hashes()).all()


@pytest.mark.parametrize("fid", [3, 1])
def test_three_pass_tables_do_not_fit(oracle, fid):
    """the three-pass plan's first pack is ~2.3 x one row; when the device cannot hold it the context falls back to the general
    kernel's plan instead of failing (LCPC_TEST_FAIL=3pass -- read only by the test-hooks build of the library,
    common.run_with_test_hooks -- simulates the failed allocation after the sub-sampled tables were made, so the clean-up runs):
    same commitment, and the context made afterwards without the hook is unaffected."""
    from common import run_with_test_hooks
    out = run_with_test_hooks("""
fid = %d
log_n = 21
n_cols, n_per_row = 1 << log_n, 1 << (log_n - 1)
coeffs = O.random_elems(fid, n_per_row + 1000, 9 + fid)
os.environ["LCPC_TEST_FAIL"] = "3pass"
enc_f = LigeroEncoding.new_from_dims(fid, n_per_row, n_cols)
del os.environ["LCPC_TEST_FAIL"]
enc = LigeroEncoding.new_from_dims(fid, n_per_row, n_cols)
a, b = LcCommit.commit(coeffs, enc_f), LcCommit.commit(coeffs, enc)
assert a.get_root() == b.get_root() and (a.comm() == b.comm()).all()
assert a.get_root() == O.Commit.commit(coeffs, O.Encoding.ligero_from_dims(fid, n_per_row, n_cols), n_threads=8).get_root()
print("fallback ok")
""" % fid)
    assert "fallback ok" in out
