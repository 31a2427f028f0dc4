"""Generates tests/golden/*.json from oracle/pyref.py (the pure-Python bignum restatement).

    python tests/golden/make_golden.py

The reference (Rust) cannot be run in this image, so these vectors pin the *restated* algorithm, not the
reference binary (see the PARITY STATUS note in oracle/pyref.py); both the C oracle (CPU tests) and the HIP
path (GPU tests) must reproduce them bit-for-bit.  Inputs are deterministic: c_i = (i+1) mod p for the
"iota" cases, ChaCha20(from_seed([s;32])) + Field::random for the "rand" cases.
"""
import hashlib
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(HERE)), "oracle"))
import pyref as P  # noqa: E402


def mont_bytes(F, vals):
    return b"".join(F.to_mont(v).to_bytes(F.nbytes, "little") for v in vals)


def coeffs_for(F, n, kind, seed):
    if kind == "iota":
        return [(i + 1) % F.p for i in range(n)]
    g = P.ChaCha20Rng(bytes([seed]) * 32)
    return [F.random(g) for _ in range(n)]


def mk_tr(root, n_col_opens):
    tr = P.Transcript(b"test transcript")
    tr.append_message(b"polycommit", root)
    tr.append_message(b"ncols", n_col_opens.to_bytes(8, "big"))
    return tr


def commit_bincode(F, c):
    """bincode 1.3 of WrappedLcCommit (lcpc-2d/src/lib.rs:186-197): Vec<F> = u64 len + raw Montgomery limbs, usize = u64,
    Vec<WrappedOutput> = u64 len + (u64 32 + digest) each"""
    padded = list(c.coeffs) + [0] * (c.n_rows * c.n_per_row - len(c.coeffs))
    out = [len(c.comm).to_bytes(8, "little"), mont_bytes(F, c.comm), len(padded).to_bytes(8, "little"), mont_bytes(F, padded)]
    out += [v.to_bytes(8, "little") for v in (c.n_rows, c.n_cols, c.n_per_row)]
    out.append(len(c.hashes).to_bytes(8, "little"))
    out += [(32).to_bytes(8, "little") + h for h in c.hashes]
    return b"".join(out)


def commit_case(name, F, enc, enc_desc, n, kind, seed, with_proof=True):
    coeffs = coeffs_for(F, n, kind, seed)
    c = P.commit(F, coeffs, enc)
    ser_c = commit_bincode(F, c)
    out = dict(name=name, field=F.fid, enc=enc_desc, n_coeffs=n, coeffs=kind, seed=seed,
               n_rows=c.n_rows, n_per_row=c.n_per_row, n_cols=c.n_cols,
               n_col_opens=enc.get_n_col_opens(), n_degree_tests=enc.get_n_degree_tests(),
               root=c.get_root().hex(),
               comm_sha256=hashlib.sha256(mont_bytes(F, c.comm)).hexdigest(),
               hashes_sha256=hashlib.sha256(b"".join(c.hashes)).hexdigest(),
               comm_head=[hex(F.to_mont(v)) for v in c.comm[:4]],
               comm_row0_col1_repr=F.to_repr(c.comm[1]).hex(),
               leaf0=c.hashes[0].hex(),
               commit_bincode_len=len(ser_c), commit_bincode_sha256=hashlib.sha256(ser_c).hexdigest())
    if with_proof:
        x = 0x1234567 % F.p
        inner = [pow(x, i, F.p) for i in range(c.n_per_row)]
        xr = pow(x, c.n_per_row, F.p)
        outer = [pow(xr, i, F.p) for i in range(c.n_rows)]
        pf, cols = P.prove(F, c, outer, enc, mk_tr(c.get_root(), enc.get_n_col_opens()))
        ser = P.ser_proof(F, pf)
        ev = P.verify(F, c.get_root(), outer, inner, pf, enc, mk_tr(c.get_root(), enc.get_n_col_opens()))
        assert ev == sum(cf * pow(x, i, F.p) for i, cf in enumerate(coeffs)) % F.p
        out.update(eval_point=hex(x), proof_len=len(ser), proof_sha256=hashlib.sha256(ser).hexdigest(),
                   proof_blake3=P.blake3(ser).hex(),
                   cols_opened_head=cols[:8], eval=hex(ev),
                   p_eval_head=[hex(F.to_mont(v)) for v in pf.p_eval[:2]])
    return out


def main():
    cases = []
    cases.append(commit_case("ligero_ft63_2e10_iota", P.FT63, P.LigeroEncoding.new(P.FT63, 1024),
                             dict(kind="ligero", rho=[1, 2], length=1024), 1024, "iota", 0))
    cases.append(commit_case("ligero_ft63_1000_rand", P.FT63, P.LigeroEncoding.new(P.FT63, 1000),
                             dict(kind="ligero", rho=[1, 2], length=1000), 1000, "rand", 1))
    cases.append(commit_case("ligero_ft127_777_rho14", P.FT127, P.LigeroEncoding.new(P.FT127, 777, (1, 4)),
                             dict(kind="ligero", rho=[1, 4], length=777), 777, "rand", 2))
    cases.append(commit_case("ligero_ft191_dims_100_256", P.FT191, P.LigeroEncoding(P.FT191, 100, 256, (38, 39)),
                             dict(kind="ligero", rho=[38, 39], n_per_row=100, n_cols=256), 950, "rand", 3))
    cases.append(commit_case("ligero_ft255_2e12_iota", P.FT255, P.LigeroEncoding.new(P.FT255, 4096),
                             dict(kind="ligero", rho=[1, 2], length=4096), 4096, "iota", 0))
    cases.append(commit_case("ligero_ft255_3000_rand", P.FT255, P.LigeroEncoding.new(P.FT255, 3000),
                             dict(kind="ligero", rho=[1, 2], length=3000), 3000, "rand", 4))
    cases.append(commit_case("sdig_ft255_600_seed0", P.FT255, P.SdigEncoding.new(P.FT255, 600, 0),
                             dict(kind="sdig", code=3, seed=0, length=600), 600, "rand", 5))
    cases.append(commit_case("sdig_ft63_900_code5", P.FT63, P.SdigEncoding.new(P.FT63, 900, 77, 5),
                             dict(kind="sdig", code=5, seed=77, length=900), 900, "rand", 6))
    cases.append(commit_case("sdig_ft127_2000_seed9", P.FT127, P.SdigEncoding.new(P.FT127, 2000, 9),
                             dict(kind="sdig", code=3, seed=9, length=2000), 2000, "iota", 0, with_proof=False))
    with open(os.path.join(HERE, "commit_cases.json"), "w") as f:
        json.dump(cases, f, indent=1)

    ntt = []
    for F in (P.FT63, P.FT255):
        for lg in (3, 6, 12):
            for kind in ("iota", "rand"):
                x = coeffs_for(F, 1 << lg, kind, 7)
                y = P.fft_io(F, list(x))
                ntt.append(dict(field=F.fid, log_n=lg, input=kind, seed=7,
                                out_sha256=hashlib.sha256(mont_bytes(F, y)).hexdigest(),
                                out_head=[hex(F.to_mont(v)) for v in y[:4]]))
    with open(os.path.join(HERE, "ntt_vectors.json"), "w") as f:
        json.dump(ntt, f, indent=1)

    fld = []
    for F in P.FIELDS:
        g = P.ChaCha20Rng(bytes([9]) * 32)
        tup = []
        for _ in range(16):
            a, b = F.random(g), F.random(g)
            tup.append([hex(F.to_mont(v)) for v in (a, b, a * b % F.p, (a + b) % F.p, (a - b) % F.p)] +
                       [F.to_repr(a).hex()])
        fld.append(dict(field=F.fid, name=F.name, modulus=hex(F.p), R=hex(F.R), R2=hex(F.R2), inv64=hex(F.inv64),
                        root_of_unity=hex(F.root_of_unity), S=F.S, L=F.L, tuples=tup))
    with open(os.path.join(HERE, "field_kats.json"), "w") as f:
        json.dump(fld, f, indent=1)

    b3 = []
    for n in (288, 3264, 8224, 16416, 32800):     # leaf-message lengths of C1, C3, C2, headline, C4
        msg = b"\0" * 32 + bytes((7 * i + 3) % 256 for i in range(n - 32))
        b3.append(dict(len=n, digest=P.blake3(msg).hex()))
    with open(os.path.join(HERE, "blake3_leaf_shapes.json"), "w") as f:
        json.dump(b3, f, indent=1)
    # shapes chosen by the constructors: new(len) for the BASELINE.json lengths and new_ml(n_vars) (ligero lib.rs:128-135,
    # brakedown lib.rs:114-123); null where the reference's assert!s reject the multilinear split
    dims = []
    for F in (P.FT63, P.FT255):
        for lg in (10, 16, 20, 24, 26, 28):
            for rho in ((1, 2), (1, 4)):
                dims.append(dict(field=F.fid, enc="ligero", rho=list(rho), log_len=lg,
                                 new=list(P.LigeroEncoding.get_dims_len(F, 1 << lg, rho)),
                                 new_ml=(lambda d: list(d) if d else None)(P.LigeroEncoding.dims_ml(F, lg, rho))))
            for code in (1, 3, 6):
                dims.append(dict(field=F.fid, enc="sdig", code=code, log_len=lg,
                                 new=list(P.SdigEncoding.dims_only(F, 1 << lg, code)),
                                 new_ml=list(P.SdigEncoding.dims_only(F, 1 << lg, code, ml=True))))
    with open(os.path.join(HERE, "constructor_dims.json"), "w") as f:
        json.dump(dims, f, indent=1)
    print("wrote golden fixtures:", len(cases), "commit cases")


if __name__ == "__main__":
    main()
