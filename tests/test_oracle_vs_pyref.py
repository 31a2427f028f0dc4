"""C oracle (oracle/lcpc_oracle.c) vs the independent bignum restatement (oracle/pyref.py) on small,
seeded inputs; plus the relations the reference's own tests assert (lcpc-2d/src/tests.rs:127-420,
lcpc-ligero-pc/src/tests.rs:22-41, 216-399; lcpc-brakedown-pc/src/tests.rs:192-375).  CPU only."""
import ctypes as C
import random

import numpy as np
import pytest

import pyref as P


def test_log2():
    # lcpc-2d/src/tests.rs:127-134
    for k in range(31):
        assert P.log2_ceil(1 << k) == k


@pytest.mark.parametrize("F", P.FIELDS, ids=lambda f: f.name)
def test_field_arith_ntt_random(oracle, F):
    O = oracle
    rnd = random.Random(7 + F.fid)
    fi = O.field_info(F.fid)
    assert fi["L"] == F.L and fi["S"] == F.S and fi["num_bits"] == F.num_bits and fi["inv"] == F.inv64
    assert O.limbs_to_ints(fi["r"][None, :])[0] == F.R
    assert O.limbs_to_ints(fi["r2"][None, :])[0] == F.R2
    assert O.limbs_to_ints(fi["rou"][None, :])[0] == F.to_mont(F.root_of_unity)
    a = [rnd.randrange(F.p) for _ in range(50)] + [0, 1, F.p - 1]
    b = [rnd.randrange(F.p) for _ in range(50)] + [F.p - 1, F.p - 1, F.p - 1]
    am, bm = O.to_mont(F.fid, a), O.to_mont(F.fid, b)
    assert O.limbs_to_ints(am) == [F.to_mont(x) for x in a]
    o = np.zeros_like(am)
    O.lib().lo_f_mul(F.fid, O.ptr(am), O.ptr(bm), O.ptr(o), len(a))
    assert O.to_canon_ints(F.fid, o) == [x * y % F.p for x, y in zip(a, b)]
    O.lib().lo_f_add(F.fid, O.ptr(am), O.ptr(bm), O.ptr(o), len(a))
    assert O.to_canon_ints(F.fid, o) == [(x + y) % F.p for x, y in zip(a, b)]
    O.lib().lo_f_sub(F.fid, O.ptr(am), O.ptr(bm), O.ptr(o), len(a))
    assert O.to_canon_ints(F.fid, o) == [(x - y) % F.p for x, y in zip(a, b)]
    for lg in (0, 1, 3, 6, 8):
        x = [rnd.randrange(F.p) for _ in range(1 << lg)]
        xm = O.to_mont(F.fid, x)
        O.lib().lo_fft_io(F.fid, O.ptr(xm), lg)
        got = O.to_canon_ints(F.fid, xm)
        assert got == P.fft_io(F, list(x))
        if lg <= 6:
            # defining property: out[bitrev(k)] = sum_i x_i w^(ik)
            w = pow(F.root_of_unity, 1 << (F.S - lg), F.p)
            for k in range(1 << lg):
                assert got[P.bitrev(k, lg)] == sum(xi * pow(w, i * k, F.p) for i, xi in enumerate(x)) % F.p
    g = P.ChaCha20Rng(bytes([5]) * 32)
    assert O.to_canon_ints(F.fid, O.random_elems(F.fid, 20, 5)) == [F.random(g) for _ in range(20)]


def test_transcript_and_rng_streams(oracle):
    O = oracle
    rnd = random.Random(3)
    t1, t2 = O.Transcript(b"x"), P.Transcript(b"x")
    for _ in range(40):
        m = bytes(rnd.randrange(256) for _ in range(rnd.randrange(0, 400)))
        t1.append_message(b"lbl", m)
        t2.append_message(b"lbl", m)
        k = rnd.randrange(1, 300)
        assert t1.challenge_bytes(b"ch", k) == t2.challenge_bytes(b"ch", k)
    g1 = O.lib().lo_rng_seed_from_u64(12345)
    O.lib().lo_rng_set_stream(g1, 3)
    g2 = P.ChaCha20Rng.seed_from_u64(12345)
    g2.set_stream(3)
    for i in range(300):
        if i % 3 == 0:
            assert O.lib().lo_rng_next_u32(g1) == g2.next_u32()
        elif i % 3 == 1:
            assert O.lib().lo_rng_next_u64(g1) == g2.next_u64()
        else:
            h = rnd.randrange(1, 1 << 40)
            assert O.lib().lo_rng_uniform(g1, h) == P.uniform_usize(g2, h)
    O.lib().lo_rng_free(g1)


def test_get_dims_property(oracle):
    # lcpc-ligero-pc/src/tests.rs:22-41 + C-vs-py agreement for ligero and all six SDIG codes
    O = oracle
    rnd = random.Random(11)
    for F in (P.FT63, P.FT255):
        for rho in ((1, 2), (1, 4), (38, 39)):
            for _ in range(150):
                lgl = 8 + rnd.randrange(8) if rnd.random() < 0.5 else rnd.randrange(2, 30)
                n = (1 << (lgl - 1)) + rnd.randrange(1 << (lgl - 1))
                a, b, c = C.c_uint64(), C.c_uint64(), C.c_uint64()
                assert O.lib().lo_ligero_get_dims(F.fid, n, rho[0], rho[1], C.byref(a), C.byref(b), C.byref(c)) == 0
                nr, np_, nc = a.value, b.value, c.value
                assert (nr, np_, nc) == P.LigeroEncoding.get_dims_len(F, n, rho)
                assert nr * np_ >= n > (nr - 1) * np_
                assert np_ * rho[1] // rho[0] <= nc and np_ < nc and nc & (nc - 1) == 0
        for code in range(1, 7):
            for _ in range(40):
                n = rnd.randrange(50, 1 << rnd.randrange(7, 30))
                a, b, c = C.c_uint64(), C.c_uint64(), C.c_uint64()
                assert O.lib().lo_sdig_get_dims(F.fid, n, code, C.byref(a), C.byref(b), C.byref(c)) == 0
                assert (a.value, b.value, c.value) == P.SdigEncoding.dims_only(F, n, code)


def test_new_ml_dims(oracle):
    """new_ml (ligero lib.rs:128-135, brakedown lib.rs:114-123): C oracle == pyref for every n_vars, all rates / codes;
    the Ligero assert!s (power-of-two split of 2^n_vars) fire in both or in neither."""
    O = oracle
    for F in (P.FT63, P.FT127, P.FT255):
        for n_vars in range(1, 31):
            a, b, c = C.c_uint64(), C.c_uint64(), C.c_uint64()
            for rho in ((1, 2), (1, 4), (38, 39)):
                rc = O.lib().lo_ligero_get_dims_ml(F.fid, n_vars, rho[0], rho[1], C.byref(a), C.byref(b), C.byref(c))
                want = P.LigeroEncoding.dims_ml(F, n_vars, rho)
                assert (rc == 0) == (want is not None), (F.fid, n_vars, rho)
                if want:
                    assert (a.value, b.value, c.value) == want
                    assert a.value * b.value == 1 << n_vars
            for code in range(1, 7):
                if (1 << n_vars) < 64:
                    continue
                assert O.lib().lo_sdig_get_dims_ml(F.fid, n_vars, code, C.byref(a), C.byref(b), C.byref(c)) == 0
                assert (a.value, b.value, c.value) == P.SdigEncoding.dims_only(F, 1 << n_vars, code, ml=True)
                assert b.value & (b.value - 1) == 0 or b.value == 1 << n_vars       # n_per_row: a power of two


def _mk_tr(T, root, n_col_opens):
    # lcpc-ligero-pc/src/tests.rs:243-245
    tr = T(b"test transcript")
    tr.append_message(b"polycommit", root)
    tr.append_message(b"ncols", n_col_opens.to_bytes(8, "big"))
    return tr


CASES = [
    ("ligero-ft63", P.FT63, lambda: P.LigeroEncoding.new(P.FT63, 1000), lambda O: O.Encoding.ligero(0, 1000), 1000),
    ("ligero-ft255", P.FT255, lambda: P.LigeroEncoding.new(P.FT255, 3000), lambda O: O.Encoding.ligero(3, 3000), 3000),
    ("ligero-ft127-rho1/4", P.FT127, lambda: P.LigeroEncoding.new(P.FT127, 777, (1, 4)),
     lambda O: O.Encoding.ligero(1, 777, (1, 4)), 777),
    ("ligero-ft191-dims", P.FT191, lambda: P.LigeroEncoding(P.FT191, 100, 256, (38, 39)),
     lambda O: O.Encoding.ligero_from_dims(2, 100, 256, (38, 39)), 950),
    ("sdig-ft255", P.FT255, lambda: P.SdigEncoding.new(P.FT255, 600, 0), lambda O: O.Encoding.sdig(3, 600, 0), 600),
    ("sdig-ft63-code5", P.FT63, lambda: P.SdigEncoding.new(P.FT63, 900, 77, 5), lambda O: O.Encoding.sdig(0, 900, 77, 5), 900),
]


@pytest.mark.parametrize("name,F,mk_py,mk_c,n", CASES, ids=[c[0] for c in CASES])
def test_end_to_end_c_vs_py(oracle, name, F, mk_py, mk_c, n):
    """commit -> prove -> bincode -> verify (lcpc-2d/src/tests.rs:238-321), C oracle == pyref bit-for-bit."""
    O = oracle
    rnd = random.Random(hash(name) & 0xFFFF)
    coeffs = [rnd.randrange(F.p) for _ in range(n)]
    pe, ce = mk_py(), mk_c(O)
    assert pe.get_dims(n) == ce.get_dims(n)
    assert pe.get_n_col_opens() == ce.get_n_col_opens() and pe.get_n_degree_tests() == ce.get_n_degree_tests()
    pc = P.commit(F, coeffs, pe)
    cc = O.Commit.commit(O.to_mont(F.fid, coeffs), ce, n_threads=2)
    assert O.to_canon_ints(F.fid, cc.comm()) == pc.comm
    assert [bytes(h) for h in cc.hashes()] == pc.hashes
    assert cc.get_root() == pc.get_root()
    # merkleize (parallel) == merkleize_ser (lcpc-2d/src/tests.rs:136-149)
    cc2 = O.Commit.from_parts(ce, cc.comm(), cc.coeffs(), cc.n_rows)
    cc2.merkleize_ser()
    assert (cc2.hashes() == cc.hashes()).all()
    x = rnd.randrange(F.p)
    inner = [pow(x, i, F.p) for i in range(pc.n_per_row)]
    xr = pow(x, pc.n_per_row, F.p)
    outer = [pow(xr, i, F.p) for i in range(pc.n_rows)]
    ev = sum(cf * pow(x, i, F.p) for i, cf in enumerate(coeffs)) % F.p
    root = pc.get_root()
    nco = pe.get_n_col_opens()
    # eval_outer == eval_outer_ser (tests.rs:151-165)
    assert O.to_canon_ints(F.fid, cc.collapse(O.to_mont(F.fid, outer), n_threads=2)) == \
        P.collapse_columns(F, pc.coeffs, outer, pc.n_rows, pc.n_per_row)
    pf, cols = P.prove(F, pc, outer, pe, _mk_tr(P.Transcript, root, nco))
    cpf, ccols = cc.prove(O.to_mont(F.fid, outer), ce, _mk_tr(O.Transcript, root, nco))
    assert list(ccols) == cols
    assert cpf == P.ser_proof(F, pf)
    assert len(cpf) == P.proof_size(F, pc.n_rows, pc.n_per_row, pc.n_cols, nco, pe.get_n_degree_tests())
    assert P.verify(F, root, outer, inner, pf, pe, _mk_tr(P.Transcript, root, nco)) == ev
    rc, evl = O.verify(ce, root, O.to_mont(F.fid, outer), O.to_mont(F.fid, inner), cpf, _mk_tr(O.Transcript, root, nco))
    assert rc == 0 and O.to_canon_ints(F.fid, evl[None, :]) == [ev]
    bad = bytearray(cpf)
    bad[len(bad) // 2] ^= 1
    rc, _ = O.verify(ce, root, O.to_mont(F.fid, outer), O.to_mont(F.fid, inner), bytes(bad), _mk_tr(O.Transcript, root, nco))
    assert rc != 0
    rc, _ = O.verify(ce, bytes(32), O.to_mont(F.fid, outer), O.to_mont(F.fid, inner), cpf, _mk_tr(O.Transcript, root, nco))
    assert rc == -33  # ColumnPath


def test_open_column_roundtrip(oracle):
    # lcpc-2d/src/tests.rs:167-191: random comm, merkleize, open 64 columns, check the path against the root
    O = oracle
    rnd = random.Random(5)
    enc = O.Encoding.ligero_from_dims(0, 100, 256)
    n_rows = 7
    comm = O.random_elems(0, n_rows * 256, 9)
    c = O.Commit.from_parts(enc, comm, None, n_rows)
    c.merkleize(2)
    root = c.get_root()
    for _ in range(64):
        col = rnd.randrange(256)
        cv, path = c.open_column(col)
        assert (cv == comm.reshape(n_rows, 256, -1)[:, col]).all()
        h = O.hash_column(0, cv)
        cn = col
        for p in path:
            h = P.blake3(h + bytes(p)) if cn % 2 == 0 else P.blake3(bytes(p) + h)
            cn >>= 1
        assert h == root
    with pytest.raises(RuntimeError):
        c.open_column(256)      # ProverError::ColumnNumber (lib.rs:797-799)


def test_commit_is_codeword(oracle):
    # lcpc-2d/src/tests.rs:193-236: RLC of encoded rows -> ifft_oi -> zero high part, matches eval_outer
    O, F = oracle, P.FT63
    rnd = random.Random(9)
    n = 700
    coeffs = [rnd.randrange(F.p) for _ in range(n)]
    enc = O.Encoding.ligero_from_dims(0, 40, 64)       # non-power-of-two n_per_row (rho random in the reference test)
    c = O.Commit.commit(O.to_mont(0, coeffs), enc)
    comm = np.array(O.to_canon_ints(0, c.comm()), dtype=object).reshape(c.n_rows, c.n_cols)
    tensor = [rnd.randrange(F.p) for _ in range(c.n_rows)]
    rlc = [int(sum(int(comm[r][j]) * tensor[r] for r in range(c.n_rows)) % F.p) for j in range(c.n_cols)]
    nat = P.ifft_oi(F, rlc)
    assert all(v == 0 for v in nat[c.n_per_row:])
    assert nat[:c.n_per_row] == O.to_canon_ints(0, c.collapse(O.to_mont(0, tensor)))


def test_sdig_matrices_match_py(oracle):
    # matgen.rs:28-188: same (seed, n) -> same CSC matrices in both restatements
    O, F = oracle, P.FT127
    pe = P.SdigEncoding(F, 500, seed=42, code=3)
    ce = O.Encoding.sdig_from_dims(F.fid, 500, 0, 42, 3)
    mats = ce.sdig_matrices()
    assert len(mats) == len(pe.pre)
    for lev, (cpre, cpost) in enumerate(mats):
        for cm, pm in ((cpre, pe.pre[lev]), (cpost, pe.post[lev])):
            rows, cols, colptr, rowidx, vals = cm
            assert (rows, cols) == pm[0]
            canon = O.to_canon_ints(F.fid, vals)
            k = 0
            for j, col in enumerate(pm[1]):
                assert colptr[j] == k
                for (i, v) in col:
                    assert rowidx[k] == i and canon[k] == v
                    k += 1
            assert colptr[cols] == k
