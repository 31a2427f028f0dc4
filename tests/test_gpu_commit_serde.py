"""serde of LcCommit itself (lcpc-2d/src/lib.rs:186-268: WrappedLcCommit in bincode 1.3's default layout) -- the hand-off of a
whole COMMITMENT between this library and the reference: the streamed bytes must be exactly what the reference's
`Serialize for LcCommit` writes for the oracle's commitment of the same coefficients, and a commitment read back from such
bytes must prove to the same proof bytes."""
import io
import struct

import numpy as np
import pytest

import lcpc_amd
from common import commit_bincode, golden_coeffs, load_golden, mk_transcript
from lcpc_amd import LcCommit, LcpcError, LigeroEncoding, SdigEncoding, Transcript

pytestmark = pytest.mark.gpu


oracle_bincode = commit_bincode


CASES = [("ligero", 3, 1 << 20, None), ("ligero", 3, (1 << 16) - 77, None), ("ligero", 0, 1 << 16, None), ("ligero", 1, 1 << 14, (1, 4)),
         ("ligero", 2, 5000, (38, 39)), ("sdig", 3, 1 << 14, None), ("sdig", 3, 1 << 20, None), ("sdig", 1, 3000, None)]


@pytest.mark.parametrize("kind,fid,n,rho", CASES)
def test_commit_bincode_roundtrip(oracle, kind, fid, n, rho):
    O = oracle
    coeffs = O.random_elems(fid, n, 77 + fid)
    if kind == "ligero":
        kw = dict(rho=rho) if rho else {}
        enc, oenc = LigeroEncoding.new(fid, n, **kw), O.Encoding.ligero(fid, n, **kw)
    else:
        enc, oenc = SdigEncoding.new(fid, n, 5), O.Encoding.sdig(fid, n, 5)
    c = LcCommit.commit(coeffs, enc)
    oc = O.Commit.commit(coeffs, oenc)
    want = oracle_bincode(oc)
    buf = io.BytesIO()
    c.to_bincode(buf)
    assert c.bincode_size() == len(want)
    assert buf.getvalue() == want
    # ... and back: a commitment deserialised from the reference-format bytes proves to the oracle's proof bytes
    d = LcCommit.from_bincode(enc, io.BytesIO(want))
    assert d.get_root() == oc.get_root() and (d.hashes() == oc.hashes()).all()
    assert (d.comm() == oc.comm()).all() and (d.coeffs() == oc.coeffs()).all()
    outer = O.random_elems(fid, oc.n_rows, 9)
    root = oc.get_root()
    pf = d.prove(outer, enc, mk_transcript(Transcript, root, enc.get_n_col_opens()))
    opf, _ = oc.prove(outer, oenc, mk_transcript(O.Transcript, root, oenc.get_n_col_opens()))
    assert pf.to_bytes() == opf


def test_commit_bincode_rejects_bad_streams(oracle):
    O, fid, n = oracle, 3, 1 << 12
    coeffs = O.random_elems(fid, n, 5)
    enc, oenc = LigeroEncoding.new(fid, n), O.Encoding.ligero(fid, n)
    good = oracle_bincode(O.Commit.commit(coeffs, oenc))
    nr, npr, nc = enc.get_dims(n)

    def status(b):
        with pytest.raises(LcpcError) as e:
            LcCommit.from_bincode(enc, io.BytesIO(b))
        return e.value.code

    assert status(good[:len(good) - 1]) == lcpc_amd.ERR_ARG                                  # truncated stream
    bad = bytearray(good); bad[8 + 40] ^= 1                                                  # a comm element: its leaf digest no longer matches
    assert status(bytes(bad)) == lcpc_amd.ERR_COMMIT
    bad = bytearray(good); bad[-1] ^= 1                                                      # the root digest itself
    assert status(bytes(bad)) == lcpc_amd.ERR_COMMIT
    bad = bytearray(good); bad[8 + 24:8 + 32] = b"\xff" * 8                                   # a limb vector >= p
    assert status(bytes(bad)) == lcpc_amd.ERR_COMMIT
    off_dims = 8 + nr * nc * 32 + 8 + nr * npr * 32
    bad = bytearray(good); bad[off_dims + 8:off_dims + 16] = struct.pack("<Q", nc * 2)       # n_cols that is not the encoder's
    assert status(bytes(bad)) == lcpc_amd.ERR_COMMIT
    bad = bytearray(good); bad[0:8] = struct.pack("<Q", nr * nc + 1)                         # comm.len() not a multiple of n_cols
    assert status(bytes(bad)) == lcpc_amd.ERR_COMMIT
    # an untrusted length the device could not hold (2^28 rows of this encoder: below the format's 2^40-element cap, above any
    # HBM): refused before anything is freed or allocated -- a commitment held by the object survives the attempt
    held = LcCommit.commit(coeffs, enc)
    root = held.get_root()
    bad = bytearray(good); bad[0:8] = struct.pack("<Q", (1 << 28) * nc)
    err = []
    rc = lcpc_amd._lib.lib().lcpc_commit_from_bincode
    def rd(_u, data, k, src=io.BytesIO(bytes(bad))):
        b = src.read(k)
        if len(b) != k:
            return 1
        import ctypes as C
        C.memmove(data, b, k)
        return 0
    assert rc(held._h, lcpc_amd._lib.WRITE_FN(rd), None, None) == lcpc_amd.ERR_COMMIT
    assert held.get_root() == root
    # nothing is left committed after a refused stream
    cm = LcCommit(enc)
    with pytest.raises(LcpcError):
        cm.get_root()


def test_commit_bincode_mutation_sweep(oracle):
    """random single-bit flips and truncations of a serialised commitment: the reader either refuses the stream (a digest that
    no longer belongs to comm, a limb vector >= p, a length or dimension that does not fit the encoder, a short read) or -- when
    the flip landed in `coeffs`, which no digest covers (the reference's Deserialize would take it too) -- accepts it with the
    same root; it never crashes and never leaves a half-built commitment behind."""
    import random
    O, fid, n = oracle, 1, 3000
    coeffs = O.random_elems(fid, n, 15)
    enc, oenc = LigeroEncoding.new(fid, n), O.Encoding.ligero(fid, n)
    oc = O.Commit.commit(coeffs, oenc)
    good = oracle_bincode(oc)
    nr, npr, nc = oc.n_rows, oc.n_per_row, oc.n_cols
    coeffs_lo, coeffs_hi = 8 + nr * nc * 16 + 8, 8 + nr * nc * 16 + 8 + nr * npr * 16
    rnd = random.Random(7)
    accepted = refused = 0
    for i in range(60):
        bad = bytearray(good)
        if i % 6 == 5:
            bad = bad[:rnd.randrange(len(bad))]
        else:
            pos = rnd.randrange(len(bad))
            bad[pos] ^= 1 << rnd.randrange(8)
        try:
            d = LcCommit.from_bincode(enc, io.BytesIO(bytes(bad)))
        except LcpcError as e:
            assert e.code in (lcpc_amd.ERR_ARG, lcpc_amd.ERR_COMMIT), e.code
            refused += 1
            continue
        assert len(bad) == len(good) and coeffs_lo <= pos < coeffs_hi, "a mutated stream outside coeffs was accepted (byte %d)" % pos
        assert d.get_root() == oc.get_root()
        accepted += 1
    assert refused >= 40


@pytest.mark.parametrize("case", load_golden("commit_cases.json"), ids=lambda c: c["name"])
def test_commit_bincode_matches_golden(oracle, case):
    """the streamed serde bytes of every golden commitment: length and sha256 as committed in tests/golden/commit_cases.json
    (made by the Python restatement; oracle/repin prints the same keys from the real crates)"""
    import hashlib
    O = oracle
    e = case["enc"]
    fid = case["field"]
    if e["kind"] == "ligero":
        enc = LigeroEncoding.new(fid, e["length"], rho=tuple(e["rho"])) if "length" in e else \
            LigeroEncoding.new_from_dims(fid, e["n_per_row"], e["n_cols"], rho=tuple(e["rho"]))
    else:
        enc = SdigEncoding.new(fid, e["length"], e["seed"], e["code"])
    c = LcCommit.commit(golden_coeffs(O, case), enc)
    buf = io.BytesIO()
    c.to_bincode(buf)
    assert len(buf.getvalue()) == case["commit_bincode_len"] == c.bincode_size()
    assert hashlib.sha256(buf.getvalue()).hexdigest() == case["commit_bincode_sha256"]

