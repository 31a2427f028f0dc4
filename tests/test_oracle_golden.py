"""The C oracle reproduces every committed golden vector (tests/golden/*.json, made by
tests/golden/make_golden.py from the independent bignum restatement).  CPU only."""
import hashlib

import numpy as np
import pytest

import pyref as P
from common import commit_bincode, golden_coeffs, hex_to_limbs, load_golden, mk_transcript, powers, sha

CASES = load_golden("commit_cases.json")


def make_oracle_enc(O, case):
    e, fid = case["enc"], case["field"]
    if e["kind"] == "ligero":
        if "length" in e:
            return O.Encoding.ligero(fid, e["length"], tuple(e["rho"]))
        return O.Encoding.ligero_from_dims(fid, e["n_per_row"], e["n_cols"], tuple(e["rho"]))
    return O.Encoding.sdig(fid, e["length"], e["seed"], e["code"])


@pytest.mark.parametrize("case", CASES, ids=[c["name"] for c in CASES])
def test_commit_cases(oracle, case):
    O = oracle
    fid = case["field"]
    L = O.limbs(fid)
    enc = make_oracle_enc(O, case)
    assert enc.get_dims(case["n_coeffs"]) == (case["n_rows"], case["n_per_row"], case["n_cols"])
    assert enc.get_n_col_opens() == case["n_col_opens"] and enc.get_n_degree_tests() == case["n_degree_tests"]
    c = O.Commit.commit(golden_coeffs(O, case), enc, n_threads=2)
    assert c.get_root().hex() == case["root"]
    assert sha(c.comm()) == case["comm_sha256"]
    assert sha(c.hashes()) == case["hashes_sha256"]
    assert bytes(c.hashes()[0]).hex() == case["leaf0"]
    for k, h in enumerate(case["comm_head"]):
        assert (c.comm()[k] == hex_to_limbs(h, L)).all()
    ser = commit_bincode(c)          # serde of LcCommit (lib.rs:186-268): the C oracle's fields in the layout pyref wrote
    assert len(ser) == case["commit_bincode_len"] and hashlib.sha256(ser).hexdigest() == case["commit_bincode_sha256"]
    if "proof_len" in case:
        x = int(case["eval_point"], 16)
        outer = powers(O, fid, x, c.n_rows, c.n_per_row)
        inner = powers(O, fid, x, c.n_per_row)
        proof, cols = c.prove(outer, enc, mk_transcript(O.Transcript, c.get_root(), case["n_col_opens"]))
        assert len(proof) == case["proof_len"] and hashlib.sha256(proof).hexdigest() == case["proof_sha256"]
        assert list(cols[:8]) == case["cols_opened_head"]
        rc, ev = O.verify(enc, c.get_root(), outer, inner, proof, mk_transcript(O.Transcript, c.get_root(), case["n_col_opens"]))
        assert rc == 0 and O.to_canon_ints(fid, ev[None, :])[0] == int(case["eval"], 16)   # golden eval is canonical


def test_ntt_vectors(oracle):
    O = oracle
    for v in load_golden("ntt_vectors.json"):
        fid, lg = v["field"], v["log_n"]
        x = golden_coeffs(O, dict(field=fid, n_coeffs=1 << lg, coeffs=v["input"], seed=v["seed"]))
        O.lib().lo_fft_io(fid, O.ptr(x), lg)
        assert sha(x) == v["out_sha256"]
        for k, h in enumerate(v["out_head"]):
            assert (x[k] == hex_to_limbs(h, O.limbs(fid))).all()


def test_field_kats(oracle):
    O = oracle
    for f in load_golden("field_kats.json"):
        fid, L = f["field"], f["L"]
        fi = O.field_info(fid)
        assert (fi["modulus"] == hex_to_limbs(f["modulus"], L)).all()
        assert (fi["r"] == hex_to_limbs(f["R"], L)).all() and (fi["r2"] == hex_to_limbs(f["R2"], L)).all()
        assert fi["inv"] == int(f["inv64"], 16) and fi["S"] == f["S"]
        for a, b, ab, apb, amb, repr_a in f["tuples"]:
            am, bm = hex_to_limbs(a, L)[None, :].copy(), hex_to_limbs(b, L)[None, :].copy()
            o = np.zeros_like(am)
            O.lib().lo_f_mul(fid, O.ptr(am), O.ptr(bm), O.ptr(o), 1)
            assert (o[0] == hex_to_limbs(ab, L)).all()
            O.lib().lo_f_add(fid, O.ptr(am), O.ptr(bm), O.ptr(o), 1)
            assert (o[0] == hex_to_limbs(apb, L)).all()
            O.lib().lo_f_sub(fid, O.ptr(am), O.ptr(bm), O.ptr(o), 1)
            assert (o[0] == hex_to_limbs(amb, L)).all()
            rb = np.zeros(8 * L, np.uint8)
            O.lib().lo_f_to_repr(fid, O.ptr(am), O.ptr(rb), 1)
            assert rb.tobytes().hex() == repr_a


def test_blake3_leaf_shapes(oracle):
    for v in load_golden("blake3_leaf_shapes.json"):
        n = v["len"]
        msg = b"\0" * 32 + bytes((7 * i + 3) % 256 for i in range(n - 32))
        assert oracle.blake3(msg).hex() == v["digest"]


def test_constructor_dims_fixture(oracle):
    """tests/golden/constructor_dims.json: shapes picked by new(len) and new_ml(n_vars) for the BASELINE.json lengths."""
    import ctypes as C
    import json
    import os
    O = oracle
    rows = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "constructor_dims.json")))
    a, b, c = C.c_uint64(), C.c_uint64(), C.c_uint64()
    for r in rows:
        fid, lg = r["field"], r["log_len"]
        if r["enc"] == "ligero":
            rn, rd = r["rho"]
            assert O.lib().lo_ligero_get_dims(fid, 1 << lg, rn, rd, C.byref(a), C.byref(b), C.byref(c)) == 0
            assert [a.value, b.value, c.value] == r["new"]
            rc = O.lib().lo_ligero_get_dims_ml(fid, lg, rn, rd, C.byref(a), C.byref(b), C.byref(c))
            assert (rc == 0) == (r["new_ml"] is not None)
            if rc == 0:
                assert [a.value, b.value, c.value] == r["new_ml"]
        else:
            assert O.lib().lo_sdig_get_dims(fid, 1 << lg, r["code"], C.byref(a), C.byref(b), C.byref(c)) == 0
            assert [a.value, b.value, c.value] == r["new"]
            assert O.lib().lo_sdig_get_dims_ml(fid, lg, r["code"], C.byref(a), C.byref(b), C.byref(c)) == 0
            assert [a.value, b.value, c.value] == r["new_ml"]
