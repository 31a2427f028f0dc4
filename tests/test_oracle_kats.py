"""Pins the oracle (oracle/pyref.py and oracle/lcpc_oracle.c) against every fixed value available:
upstream published vectors of the third-party primitives and the reference's own
36 published proof sizes (doc/benchmark-results/*_pvs.txt).  CPU only."""
import hashlib
import struct

import pyref as P

PAT = lambda n: bytes(i % 251 for i in range(n))
B3_KATS = [
    (b"", "af1349b9f5f9a1a6a0404dea36dcc9499bcb25c9adc112b7cc9a93cae41f3262"),
    (b"abc", "6437b3ac38465133ffb63b75273a8db548c558465d79db03fd359c6cd5bd9d85"),
    (PAT(1), "2d3adedff11b61f14c886e35afa036736dcd87a74d27b5c1510225d0f592e213"),
    (PAT(1025), "d00278ae47eb27b34faecf67b4fe263f82d5412916c1ffd97c8cb7fb814b8444"),
]


def test_blake3_official_vectors(oracle):
    for msg, hexd in B3_KATS:
        assert P.blake3(msg).hex() == hexd
        assert oracle.blake3(msg).hex() == hexd


def test_blake3_c_vs_py_lengths(oracle):
    # chunk / block boundaries and the leaf-message lengths of C1..C4
    for n in (63, 64, 65, 288, 1023, 1024, 1025, 2048, 2049, 3072, 3264, 4097, 8224, 16416, 32800):
        assert oracle.blake3(PAT(n)) == P.blake3(PAT(n)), n


def test_keccak_matches_sha3(oracle):
    import numpy as np
    st = bytearray(200)
    st[0] ^= 0x06
    st[135] ^= 0x80
    assert bytes(P.keccak_f1600(st)[:32]) == hashlib.sha3_256(b"").digest()
    a = np.frombuffer(bytes(st), np.uint8).copy()
    oracle.lib().lo_keccak_f1600(oracle.ptr(a))
    assert a[:32].tobytes() == hashlib.sha3_256(b"").digest()


def test_merlin_upstream_vector(oracle):
    # merlin 2.0 src/transcript.rs test "equivalence_simple"
    exp = "d5a21972d0d5fe320c0d263fac7fffb8145aa640af6e9bca177c03c7efcf0615"
    t = P.Transcript(b"test protocol")
    t.append_message(b"some label", b"some data")
    assert t.challenge_bytes(b"challenge", 32).hex() == exp
    t = oracle.Transcript(b"test protocol")
    t.append_message(b"some label", b"some data")
    assert t.challenge_bytes(b"challenge", 32).hex() == exp


def test_chacha20_upstream_vectors(oracle):
    r = P.ChaCha20Rng(bytes(32))
    assert [r.next_u32() for _ in range(4)] == [0xade0b876, 0x903df1a0, 0xe56a5d40, 0x28bd8653]
    seed = struct.pack("<8I", 0, 0, 1, 0, 2, 0, 3, 0)       # rand_chacha test_chacha_true_values / clone test
    assert P.ChaCha20Rng(seed).next_u32() == 137206642
    import numpy as np
    g = oracle.lib().lo_rng_from_seed(oracle.ptr(np.frombuffer(seed, np.uint8).copy()))
    assert oracle.lib().lo_rng_next_u32(g) == 137206642
    oracle.lib().lo_rng_free(g)


def test_field_constants():
    # SURVEY.md App. B.1 table
    exp = {
        "ft63": (1, 41, 0x3b12c59a0be9882c, 0x2b8e9dfffffffffd, 0x46d075ffffffffff),
        "ft127": (2, 40, 0x3280b719bea9b43abb9ee4e683614688, 0x23157ed08bbe3e8101a84dfffffffffe, 0x7f2bd8ffffffffff),
        "ft191": (3, 41, 0x3003e6b741e10f0f3fa8456e7f6a989fd3e0467894e107a2,
                  0x305ae60140ca567045c6678ad9339c96892c79fffffffffd, 0xd24681ffffffffff),
        "ft255": (4, 41, 0x5425e2a66fd9cbf775273db316b7e0c89a2e5ce2899cbfc2748b4ceb2108eb11,
                  0x33870cc92365adfe04ac41f68d514d2c211870def34d419ffab61bfffffffffe, 0x02a4f1ffffffffff),
    }
    for F in P.FIELDS:
        assert (F.L, F.S, F.root_of_unity, F.R, F.inv64) == exp[F.name]
        assert pow(F.root_of_unity, 1 << F.S, F.p) == 1 and pow(F.root_of_unity, 1 << (F.S - 1), F.p) == F.p - 1


# reference-published proof sizes, Ft255 + BLAKE3, lgl = 13,15,...,29
SIZES = {
    ("ligero", (1, 2)): [285584, 442584, 744224, 1335144, 2504624, 4831224, 9472064, 18741384, 37267664],
    ("ligero", (1, 4)): [207704, 329184, 564584, 1027824, 1946744, 3777024, 7430024, 14728464, 29317784],
    ("ligero", (38, 39)): [4325992, 5296520, 6962536, 10019528, 15858472, 27481352, 50452008, 95898248, 186735784],
    ("sdig", 3): [4402016, 5322824, 6900784, 10250368, 16175624, 27762416, 50336784, 96017376, 186315104],
}


def test_36_published_proof_sizes(oracle):
    import ctypes as C
    F = P.FT255
    for (kind, par), exp in SIZES.items():
        for lgl, e in zip(range(13, 30, 2), exp):
            if kind == "ligero":
                nr, np_, nc = P.LigeroEncoding.get_dims_len(F, 1 << lgl, par)
                no = P.LigeroEncoding.n_col_opens_rho(par)
                a, b, c = C.c_uint64(), C.c_uint64(), C.c_uint64()
                assert oracle.lib().lo_ligero_get_dims(F.fid, 1 << lgl, par[0], par[1], C.byref(a), C.byref(b), C.byref(c)) == 0
            else:
                nr, np_, nc = P.SdigEncoding.dims_only(F, 1 << lgl, par)
                no = P.SdigEncoding.n_col_opens_code(par)
                a, b, c = C.c_uint64(), C.c_uint64(), C.c_uint64()
                assert oracle.lib().lo_sdig_get_dims(F.fid, 1 << lgl, par, C.byref(a), C.byref(b), C.byref(c)) == 0
            assert (a.value, b.value, c.value) == (nr, np_, nc)
            nd = P.n_degree_tests(128, nc, F.flog2)
            assert P.proof_size(F, nr, np_, nc, no, nd) == e, (kind, par, lgl)


def test_headline_dims():
    # SURVEY.md 8 config table / App. C
    F = P.FT255
    assert P.LigeroEncoding.get_dims_len(P.FT63, 1 << 16) == (32, 2048, 4096)
    assert P.LigeroEncoding.get_dims_len(F, 1 << 24) == (256, 65536, 131072)
    assert P.LigeroEncoding.get_dims_len(F, 1 << 26) == (512, 131072, 262144)
    assert P.LigeroEncoding.get_dims_len(F, 1 << 28) == (1024, 262144, 524288)
    assert P.SdigEncoding.dims_only(F, 1 << 24, 3) == (101, 166292, 252931)
    assert P.LigeroEncoding.n_col_opens_rho((1, 2)) == 309 and P.SdigEncoding.n_col_opens_code(3) == 6593


def test_field_definitions_independent():
    """The four field definitions (decimal modulus + generator, lcpc-test-fields/src/lib.rs:18-58) checked with sympy,
    independently of the restatement: the modulus is prime, S is the 2-adicity of p - 1, ROOT_OF_UNITY = g^t with
    p - 1 = 2^S t (ff_derive's rule [3P]) has order exactly 2^S, R = 2^(64 L) mod p, INV = -p^-1 mod 2^64."""
    import sympy
    defs = {
        "ft63": ("5102708120182849537", 10),
        "ft127": ("146823888364060453008360742206866194433", 3),
        "ft191": ("1697146272512170708389931801544665676545308500647389167617", 5),
        "ft255": ("46242760681095663677370860714659204618859642560429202607213929836750194081793", 5),
    }
    for F in P.FIELDS:
        p, g = int(defs[F.name][0]), defs[F.name][1]
        assert F.p == p and sympy.isprime(p)
        s, t = 0, p - 1
        while t % 2 == 0:
            s, t = s + 1, t // 2
        assert F.S == s
        assert pow(g, (p - 1) // 2, p) == p - 1                      # g is a quadratic non-residue: g^t has full 2-power order
        assert F.root_of_unity == pow(g, t, p)
        assert F.R == pow(2, 64 * F.L, p) and F.R2 == pow(2, 128 * F.L, p)
        assert (F.inv64 * p + 1) % (1 << 64) == 0
        assert F.flog2 == p.bit_length() - 1
