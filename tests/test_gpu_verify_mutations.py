"""The verifier on proofs it must not trust: random mutations of valid bincode proofs (bit flips in polynomials, column
values, path digests and length fields, truncations).  lcpc_verify (LcEvalProof::verify, lcpc-2d/src/lib.rs:832-952)
must neither crash nor accept, and must report the VerifierError the oracle's restatement of the reference reports for the
same bytes -- except where it is deliberately stricter: a limb vector >= p is refused as malformed,
where the reference computes on it mod p and fails later (or, for a column entry, not at all if the residue matches)."""
import random
import struct

import numpy as np
import pytest

import lcpc_amd
from common import mk_transcript, powers
from lcpc_amd import LcCommit, LcEvalProof, LigeroEncoding, SdigEncoding, Transcript

pytestmark = pytest.mark.gpu


def _setup(O, kind, fid, n, seed):
    coeffs = O.random_elems(fid, n, seed)
    if kind == "ligero":
        enc, oenc = LigeroEncoding.new(fid, n), O.Encoding.ligero(fid, n)
    else:
        enc, oenc = SdigEncoding.new(fid, n, 1), O.Encoding.sdig(fid, n, 1)
    c = LcCommit.commit(coeffs, enc)
    import pyref as P
    x = 0x1234567 % P.FIELDS[fid].p
    inner = powers(O, fid, x, c.n_per_row)
    outer = powers(O, fid, x, c.n_rows, c.n_per_row)
    root = c.get_root()
    nco = enc.get_n_col_opens()
    pf = c.prove(outer, enc, mk_transcript(Transcript, root, nco)).to_bytes()
    return enc, oenc, root, nco, inner, outer, bytes(pf), c


def _both(O, enc, oenc, root, nco, inner, outer, blob):
    try:                                                  # (deserialisation errors surface as VERR_MALFORMED too)
        LcEvalProof.from_bytes(blob, enc.L).verify(root, outer, inner, enc, mk_transcript(Transcript, root, nco))
        rc = 0
    except lcpc_amd.LcpcError as e:
        rc = e.code
    orc, _ = O.verify(oenc, root, outer, inner, blob, mk_transcript(O.Transcript, root, nco))
    return rc, orc


@pytest.mark.parametrize("kind,fid,n", [("ligero", 3, 1 << 12), ("ligero", 1, 1 << 11), ("sdig", 3, 1 << 11), ("ligero", 0, 1 << 10)])
def test_mutated_proofs_same_verdict_as_oracle(oracle, kind, fid, n):
    O = oracle
    enc, oenc, root, nco, inner, outer, pf, c = _setup(O, kind, fid, n, 5 + fid)
    assert _both(O, enc, oenc, root, nco, inner, outer, pf) == (0, 0)
    F = 8 * enc.L
    npr, n_rows = c.n_per_row, c.n_rows
    # wire layout (lib.rs:550-609): n_cols, len, p_eval, n_deg, (len, p_random)*, n_columns, (len, col, path_len, (32, digest)*)*
    off_eval = 16
    off_nd = off_eval + npr * F
    n_deg = struct.unpack_from("<Q", pf, off_nd)[0]
    off_rand = off_nd + 8 + 8
    off_ncol = off_nd + 8 + n_deg * (8 + npr * F)
    off_col0 = off_ncol + 8
    rnd = random.Random(1000 + fid)
    spots = {
        "n_cols": 0, "p_eval len": 8, "p_eval": off_eval + rnd.randrange(npr * F), "n_deg": off_nd,
        "p_random len": off_nd + 8, "p_random": off_rand + rnd.randrange(npr * F), "n_columns": off_ncol,
        "col0 len": off_col0, "col0 value": off_col0 + 8 + rnd.randrange(n_rows * F),
        "col0 path len": off_col0 + 8 + n_rows * F, "col0 digest len": off_col0 + 8 + n_rows * F + 8,
        "col0 digest": off_col0 + 8 + n_rows * F + 16 + rnd.randrange(32), "last byte": len(pf) - 1,
    }
    cases = []
    for name, pos in spots.items():
        for bit in (0, rnd.randrange(8)):
            b = bytearray(pf)
            b[pos] ^= 1 << bit
            cases.append((name + " bit %d" % bit, bytes(b)))
    for _ in range(24):                                   # anywhere
        b = bytearray(pf)
        pos = rnd.randrange(len(pf))
        b[pos] ^= 1 << rnd.randrange(8)
        cases.append(("byte %d" % pos, bytes(b)))
    cases += [("truncated", pf[:-1]), ("truncated 8", pf[:-8]), ("half", pf[:len(pf) // 2]), ("header only", pf[:16]), ("empty", b"")]
    # a limb vector >= p in p_eval: all ones in the top limb of element 0
    b = bytearray(pf)
    b[off_eval + F - 8:off_eval + F] = b"\xff" * 8
    cases.append(("p_eval[0] >= p", bytes(b)))
    n_strict = 0
    for name, blob in cases:
        rc, orc = _both(O, enc, oenc, root, nco, inner, outer, blob)
        assert rc != 0, name                              # never accepts a mutated proof
        assert orc != 0, name
        if rc == lcpc_amd.VERR_MALFORMED and orc != rc:
            n_strict += 1                                 # stricter on purpose: unreduced limbs (see the module docstring)
            continue
        assert rc == orc, (name, rc, orc)
    assert n_strict <= len(cases) // 3
    # and the untouched proof still verifies afterwards (no state left behind by the failures); trailing bytes are ignored,
    # as by bincode::deserialize (bincode 1.3's top-level functions allow them)
    assert _both(O, enc, oenc, root, nco, inner, outer, pf) == (0, 0)
    assert _both(O, enc, oenc, root, nco, inner, outer, pf + b"\0") == (0, 0)
    assert _both(O, enc, oenc, root, nco, inner, outer, pf + bytes(13)) == (0, 0)
