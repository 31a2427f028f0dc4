"""bench.py's host-side helpers that build the inputs of its `configs` legs without the oracle (so that the timed product path never
touches it): the C5 tensors (powers of x, lcpc-ligero-pc/src/tests.rs:120-128) in ff_derive's Montgomery limbs, the reference's test
transcript (tests.rs:243-245), and SURVEY.md 8(d)'s algorithmic byte count -- checked HERE against the oracle / the survey's table.  CPU only."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from common import mk_transcript, powers  # noqa: E402


def test_powers_mont_equals_oracle_conversion(oracle):
    for n, step in ((257, 1), (64, 131072), (5, 3)):
        assert (bench.powers_mont(bench.C5_X, n, step) == powers(oracle, 3, bench.C5_X, n, step)).all()



def test_bench_transcript_is_the_test_transcript(oracle):
    root = bytes(range(32))
    a = bench.mk_transcript(oracle.Transcript, root, 309)
    b = mk_transcript(oracle.Transcript, root, 309)
    assert a.challenge_bytes(b"x", 32) == b.challenge_bytes(b"x", 32)


def test_commit_bytes_follow_the_survey_formula():
    """B_commit = F nr npr + 2 F nr nc + 32 (2 np2 - 1) + 32 (2 np2 - 2).  (The survey's own table was filled in with ONE
    digest term -- its 10.754 GB at the headline is 16.8 MB below the formula's 10.771 -- the bench follows the formula as written.)"""
    for F, nr, npr, nc, table_gb in ((8, 32, 2048, 4096, 2.88e-3), (32, 256, 65536, 131072, 2.693), (32, 512, 131072, 262144, 10.754), (32, 1024, 262144, 524288, 42.98)):
        want = F * nr * npr + 2 * F * nr * nc + 32 * (2 * nc - 1) + 32 * (2 * nc - 2)
        assert bench.commit_bytes_8d(F, nr, npr, nc) == want
        assert 0 <= want / 1e9 - table_gb <= 32 * 2 * nc / 1e9 + 0.006          # the table: the same without the re-read term
    assert bench.commit_bytes_8d(32, 101, 166292, 252931) == 32 * 101 * 166292 + 2 * 32 * 101 * 252931 + 32 * (4 * (1 << 18) - 3)   # np2 of a non-power-of-two width
