"""The streaming commit of the oracle (tests/oracle_lib.py commit_streaming: row blocks -> BLAKE3 chunk chaining values -> leaf
digests -> Merkle tree, never holding comm) against the oracle's own Commit.commit, which follows lcpc-2d/src/lib.rs:622-785
statement for statement (streaming Digest per column).  The full-size GPU test at 2^28 coefficients checks the HIP path's whole
`hashes` array against the streaming form, so the streaming form must BE the commit: every `hashes` byte equal, for every field
whose elements do not straddle chunks, both encodings, ragged last rows, one chunk and many, several chunks per step.  CPU only."""
import numpy as np
import pytest


def _rows_of(coeffs, n_per_row):
    return lambda r0, r1: coeffs[r0 * n_per_row:r1 * n_per_row]


def test_streaming_commit_equals_commit_2e20_ft255(oracle):
    """2^20 Ft255 coefficients, the shape LigeroEncoding::new picks (64 x 16384 -> 32768: 3 chunks per leaf message)."""
    O, fid, n = oracle, 3, 1 << 20
    enc = O.Encoding.ligero(fid, n)
    nr, npr, nc = enc.get_dims(n)
    coeffs = O.random_elems(fid, n, 4001)
    oc = O.Commit.commit(coeffs, enc, n_threads=8)
    for step in (1, 2):
        h = O.commit_streaming(enc, n, _rows_of(coeffs, npr), n_threads=8, chunks_per_step=step)
        assert h.shape == oc.hashes().shape and (h == oc.hashes()).all()
    assert h[-1].tobytes() == oc.get_root()


@pytest.mark.parametrize("fid,n_per_row,n_cols,n_rows,short", [
    (0, 64, 128, 5, 0),            # Ft63: one chunk (32 + 5 * 8 bytes)
    (0, 64, 128, 300, 7),          # Ft63: 3 chunks, ragged last row
    (1, 32, 64, 130, 1),           # Ft127: 3 chunks
    (3, 16, 32, 31, 0),            # Ft255: exactly one full chunk (32 + 31 * 32 = 1024)
    (3, 16, 32, 32, 3),            # Ft255: one row into the second chunk
    (3, 16, 32, 200, 5),           # Ft255: 7 chunks
])
def test_streaming_commit_equals_commit_ligero_shapes(oracle, fid, n_per_row, n_cols, n_rows, short):
    O = oracle
    enc = O.Encoding.ligero_from_dims(fid, n_per_row, n_cols)
    n = n_rows * n_per_row - short
    coeffs = O.random_elems(fid, n, 4100 + n_rows)
    oc = O.Commit.commit(coeffs, enc, n_threads=2)
    for step in (1, 3, 100):
        h = O.commit_streaming(enc, n, _rows_of(coeffs, n_per_row), n_threads=3, chunks_per_step=step)
        assert (h == oc.hashes()).all(), step


def test_streaming_commit_equals_commit_brakedown(oracle):
    """a non-power-of-two number of columns: the zero padding leaves of the Merkle tree (lib.rs:656-666) are part of `hashes`."""
    O, fid, n_per_row, n_rows = oracle, 3, 900, 70
    enc = O.Encoding.sdig_from_dims(fid, n_per_row, 0, 5)
    n = n_rows * n_per_row - 11
    coeffs = O.random_elems(fid, n, 4200)
    oc = O.Commit.commit(coeffs, enc, n_threads=4)
    h = O.commit_streaming(enc, n, _rows_of(coeffs, n_per_row), n_threads=4)
    assert (h == oc.hashes()).all()


def test_streaming_commit_ft191_cuts_at_row_boundaries(oracle):
    """Ft191: 24-byte elements straddle the 1 KiB chunks; the row blocks end at rows = 84 (mod 128) only (every third chunk)"""
    O = oracle
    for n_rows, short in ((40, 0), (90, 3), (700, 5)):
        enc = O.Encoding.ligero_from_dims(2, 16, 32)
        n = n_rows * 16 - short
        coeffs = O.random_elems(2, n, 4300 + n_rows)
        oc = O.Commit.commit(coeffs, enc, n_threads=2)
        for step in (1, 2, 50):
            h = O.commit_streaming(enc, n, _rows_of(coeffs, 16), n_threads=3, chunks_per_step=step)
            assert (h == oc.hashes()).all(), (n_rows, step)
