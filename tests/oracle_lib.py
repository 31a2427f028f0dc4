"""ctypes loader for the C oracle (oracle/liblcpc_oracle.so).  Test infrastructure only:
imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg -- never by lcpc_amd/."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_DIR = os.path.join(os.path.dirname(_HERE), "oracle")
_SO = os.path.join(ORACLE_DIR, "liblcpc_oracle.so")

u64p = C.POINTER(C.c_uint64)
u8p = C.POINTER(C.c_uint8)


def build_oracle(force=False):
    src = [os.path.join(ORACLE_DIR, n) for n in ("lcpc_oracle.c", "lcpc_oracle.h")]
    if force or not os.path.exists(_SO) or any(os.path.getmtime(s) > os.path.getmtime(_SO) for s in src if os.path.exists(s)):
        subprocess.check_call(["make", "-C", ORACLE_DIR, "-s"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        build_oracle()
        L = C.CDLL(_SO)
        vp = C.c_void_p
        sigs = {
            "lo_field_limbs": (C.c_int, [C.c_int]),
            "lo_field_info": (C.c_int, [C.c_int, vp, vp, vp, vp, vp, vp, vp]),
            "lo_f_mul": (None, [C.c_int, vp, vp, vp, C.c_size_t]),
            "lo_f_add": (None, [C.c_int, vp, vp, vp, C.c_size_t]),
            "lo_f_sub": (None, [C.c_int, vp, vp, vp, C.c_size_t]),
            "lo_f_to_repr": (None, [C.c_int, vp, vp, C.c_size_t]),
            "lo_f_from_canon": (None, [C.c_int, vp, vp, C.c_size_t]),
            "lo_f_from_u64": (None, [C.c_int, vp, vp, C.c_size_t]),
            "lo_roots_table": (C.c_int, [C.c_int, C.c_uint, vp]),
            "lo_fft_io": (C.c_int, [C.c_int, vp, C.c_uint]),
            "lo_blake3": (None, [vp, C.c_size_t, vp]),
            "lo_keccak_f1600": (None, [vp]),
            "lo_rng_from_seed": (vp, [vp]),
            "lo_rng_seed_from_u64": (vp, [C.c_uint64]),
            "lo_rng_set_stream": (None, [vp, C.c_uint64]),
            "lo_rng_next_u32": (C.c_uint32, [vp]),
            "lo_rng_next_u64": (C.c_uint64, [vp]),
            "lo_rng_uniform": (C.c_uint64, [vp, C.c_uint64]),
            "lo_rng_field_random": (None, [vp, C.c_int, vp, C.c_size_t]),
            "lo_rng_free": (None, [vp]),
            "lo_tr_new": (vp, [vp, C.c_size_t]),
            "lo_tr_clone": (vp, [vp]),
            "lo_tr_append_message": (None, [vp, vp, C.c_size_t, vp, C.c_size_t]),
            "lo_tr_challenge_bytes": (None, [vp, vp, C.c_size_t, vp, C.c_size_t]),
            "lo_tr_free": (None, [vp]),
            "lo_ligero_get_dims": (C.c_int, [C.c_int, C.c_uint64, C.c_uint, C.c_uint, vp, vp, vp]),
            "lo_ligero_new": (vp, [C.c_int, C.c_uint64, C.c_uint, C.c_uint]),
            "lo_ligero_new_from_dims": (vp, [C.c_int, C.c_uint64, C.c_uint64, C.c_uint, C.c_uint]),
            "lo_sdig_get_dims": (C.c_int, [C.c_int, C.c_uint64, C.c_int, vp, vp, vp]),
            "lo_ligero_get_dims_ml": (C.c_int, [C.c_int, C.c_uint, C.c_uint, C.c_uint, vp, vp, vp]),
            "lo_sdig_get_dims_ml": (C.c_int, [C.c_int, C.c_uint, C.c_int, vp, vp, vp]),
            "lo_sdig_new": (vp, [C.c_int, C.c_uint64, C.c_uint64, C.c_int]),
            "lo_sdig_new_from_dims": (vp, [C.c_int, C.c_uint64, C.c_uint64, C.c_uint64, C.c_int]),
            "lo_enc_free": (None, [vp]),
            "lo_enc_get_dims": (None, [vp, C.c_uint64, vp, vp, vp]),
            "lo_enc_dims_ok": (C.c_int, [vp, C.c_uint64, C.c_uint64]),
            "lo_enc_n_col_opens": (C.c_uint64, [vp]),
            "lo_enc_n_degree_tests": (C.c_uint64, [vp]),
            "lo_enc_encode": (C.c_int, [vp, vp]),
            "lo_sdig_n_levels": (C.c_int, [vp]),
            "lo_sdig_matrix": (C.c_int, [vp, C.c_int, C.c_int, vp, vp, vp, vp, vp, vp]),
            "lo_commit_new": (C.c_int, [vp, vp, C.c_uint64, C.c_int, vp]),
            "lo_commit_from_parts": (C.c_int, [vp, vp, vp, C.c_uint64, vp]),
            "lo_commit_free": (None, [vp]),
            "lo_commit_dims": (None, [vp, vp, vp, vp, vp]),
            "lo_commit_comm": (vp, [vp]),
            "lo_commit_coeffs": (vp, [vp]),
            "lo_commit_hashes": (vp, [vp]),
            "lo_commit_root": (None, [vp, vp]),
            "lo_merkleize": (None, [vp, C.c_int]),
            "lo_merkleize_ser": (None, [vp]),
            "lo_collapse_columns": (C.c_int, [vp, vp, C.c_uint64, vp, C.c_int]),
            "lo_open_column": (C.c_int, [vp, C.c_uint64, vp, vp]),
            "lo_hash_column": (None, [C.c_int, vp, C.c_uint64, vp]),
            "lo_leaf_chunk_cvs": (C.c_int, [C.c_int, vp, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64, vp]),
            "lo_finish_from_cvs": (C.c_int, [vp, C.c_uint64, C.c_uint64, vp]),
            "lo_leaf_chunk_cvs_mt": (C.c_int, [C.c_int, vp, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64, vp, C.c_int]),
            "lo_finish_from_cvs_mt": (C.c_int, [vp, C.c_uint64, C.c_uint64, vp, C.c_int]),
            "lo_encode_rows": (C.c_int, [vp, vp, C.c_uint64, C.c_int, vp]),
            "lo_prove": (C.c_int, [vp, vp, vp, C.c_uint64, vp, vp, vp, vp]),
            "lo_verify": (C.c_int, [vp, vp, vp, C.c_uint64, vp, C.c_uint64, vp, C.c_uint64, vp, vp]),
            "lo_free": (None, [vp]),
        }
        for name, (res, args) in sigs.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def limbs(fid):
    return lib().lo_field_limbs(fid)


def field_info(fid):
    L = limbs(fid)
    mod, r, r2, rou = (np.zeros(L, np.uint64) for _ in range(4))
    inv = C.c_uint64()
    S, nb = C.c_uint32(), C.c_uint32()
    lib().lo_field_info(fid, ptr(mod), ptr(r), ptr(r2), C.byref(inv), ptr(rou), C.byref(S), C.byref(nb))
    return dict(L=L, modulus=mod, r=r, r2=r2, inv=inv.value, rou=rou, S=S.value, num_bits=nb.value)


def ints_to_limbs(vals, L):
    """list of python ints -> (n, L) uint64 array."""
    out = np.zeros((len(vals), L), np.uint64)
    for i, v in enumerate(vals):
        for k in range(L):
            out[i, k] = (v >> (64 * k)) & 0xFFFFFFFFFFFFFFFF
    return out


def limbs_to_ints(arr):
    arr = np.asarray(arr, np.uint64).reshape(-1, arr.shape[-1])
    return [sum(int(x) << (64 * k) for k, x in enumerate(row)) for row in arr]


def to_mont(fid, canon_ints):
    L = limbs(fid)
    c = ints_to_limbs(canon_ints, L)
    o = np.zeros_like(c)
    lib().lo_f_from_canon(fid, ptr(c), ptr(o), len(canon_ints))
    return o


def to_canon_ints(fid, mont):
    """(n, L) Montgomery limbs -> list of canonical python ints (via to_repr)."""
    mont = np.ascontiguousarray(mont, np.uint64)
    L = limbs(fid)
    n = mont.size // L
    out = np.zeros(n * 8 * L, np.uint8)
    lib().lo_f_to_repr(fid, ptr(mont), ptr(out), n)
    b = out.tobytes()
    return [int.from_bytes(b[8 * L * i:8 * L * (i + 1)], "little") for i in range(n)]


def random_elems(fid, n, seed):
    """n uniform field elements (Montgomery limbs) from ChaCha20Rng::from_seed([seed;32]) via Field::random."""
    L = limbs(fid)
    key = np.full(32, seed & 0xFF, np.uint8)
    g = lib().lo_rng_from_seed(ptr(key))
    out = np.zeros((n, L), np.uint64)
    lib().lo_rng_field_random(g, fid, ptr(out), n)
    lib().lo_rng_free(g)
    return out


def blake3(data):
    d = np.frombuffer(bytes(data), np.uint8) if len(data) else np.zeros(1, np.uint8)
    out = np.zeros(32, np.uint8)
    lib().lo_blake3(ptr(d), len(data), ptr(out))
    return out.tobytes()


class Transcript:
    def __init__(self, label=None, _h=None):
        self.h = _h if _h is not None else lib().lo_tr_new(label, len(label))

    def append_message(self, label, msg):
        lib().lo_tr_append_message(self.h, label, len(label), msg, len(msg))

    def challenge_bytes(self, label, n):
        out = C.create_string_buffer(n)
        lib().lo_tr_challenge_bytes(self.h, label, len(label), out, n)
        return out.raw

    def clone(self):
        return Transcript(_h=lib().lo_tr_clone(self.h))

    def __del__(self):
        try:
            lib().lo_tr_free(self.h)
        except Exception:
            pass


class Encoding:
    def __init__(self, h, fid):
        if not h:
            raise ValueError("oracle: encoding construction failed")
        self.h, self.fid, self.L = h, fid, limbs(fid)

    @classmethod
    def ligero(cls, fid, length, rho=(1, 2)):
        return cls(lib().lo_ligero_new(fid, length, rho[0], rho[1]), fid)

    @classmethod
    def ligero_from_dims(cls, fid, n_per_row, n_cols, rho=(1, 2)):
        return cls(lib().lo_ligero_new_from_dims(fid, n_per_row, n_cols, rho[0], rho[1]), fid)

    @classmethod
    def sdig(cls, fid, length, seed=0, code=3):
        return cls(lib().lo_sdig_new(fid, length, seed, code), fid)

    @classmethod
    def sdig_from_dims(cls, fid, n_per_row, n_cols, seed=0, code=3):
        return cls(lib().lo_sdig_new_from_dims(fid, n_per_row, n_cols, seed, code), fid)

    def get_dims(self, length):
        a, b, c = C.c_uint64(), C.c_uint64(), C.c_uint64()
        lib().lo_enc_get_dims(self.h, length, C.byref(a), C.byref(b), C.byref(c))
        return a.value, b.value, c.value

    def dims_ok(self, n_per_row, n_cols):
        return bool(lib().lo_enc_dims_ok(self.h, n_per_row, n_cols))

    def get_n_col_opens(self):
        return lib().lo_enc_n_col_opens(self.h)

    def get_n_degree_tests(self):
        return lib().lo_enc_n_degree_tests(self.h)

    def encode(self, row):
        row = np.ascontiguousarray(row, np.uint64)
        lib().lo_enc_encode(self.h, ptr(row))
        return row

    def sdig_matrices(self):
        """[(pre, post)] per level; each = (rows, cols, colptr, rowidx, vals(nnz, L))."""
        out = []
        for lev in range(lib().lo_sdig_n_levels(self.h)):
            pair = []
            for which in (0, 1):
                r, c, nnz = C.c_uint64(), C.c_uint64(), C.c_uint64()
                cp, ri, va = C.c_void_p(), C.c_void_p(), C.c_void_p()
                lib().lo_sdig_matrix(self.h, lev, which, C.byref(r), C.byref(c), C.byref(nnz),
                                     C.byref(cp), C.byref(ri), C.byref(va))
                colptr = np.ctypeslib.as_array(C.cast(cp, u64p), (c.value + 1,)).copy()
                rowidx = np.ctypeslib.as_array(C.cast(ri, u64p), (max(1, nnz.value),)).copy()[:nnz.value]
                vals = np.ctypeslib.as_array(C.cast(va, u64p), (max(1, nnz.value) * self.L,)).copy()[:nnz.value * self.L]
                pair.append((r.value, c.value, colptr, rowidx, vals.reshape(-1, self.L)))
            out.append(tuple(pair))
        return out

    def __del__(self):
        try:
            lib().lo_enc_free(self.h)
        except Exception:
            pass


class Commit:
    def __init__(self, h, enc):
        self.h, self.enc, self.L = h, enc, enc.L
        a, b, c, d = (C.c_uint64() for _ in range(4))
        lib().lo_commit_dims(h, C.byref(a), C.byref(b), C.byref(c), C.byref(d))
        self.n_rows, self.n_per_row, self.n_cols, self.n_hashes = a.value, b.value, c.value, d.value

    @classmethod
    def commit(cls, coeffs, enc, n_threads=1):
        coeffs = np.ascontiguousarray(coeffs, np.uint64)
        h = C.c_void_p()
        rc = lib().lo_commit_new(enc.h, ptr(coeffs), coeffs.size // enc.L, n_threads, C.byref(h))
        if rc:
            raise RuntimeError("oracle commit failed: %d" % rc)
        return cls(h, enc)

    @classmethod
    def from_parts(cls, enc, comm, coeffs, n_rows):
        comm = np.ascontiguousarray(comm, np.uint64)
        coeffs = np.ascontiguousarray(coeffs, np.uint64) if coeffs is not None else None
        h = C.c_void_p()
        lib().lo_commit_from_parts(enc.h, ptr(comm), ptr(coeffs) if coeffs is not None else None, n_rows, C.byref(h))
        return cls(h, enc)

    def comm(self):
        p = C.cast(lib().lo_commit_comm(self.h), u64p)
        return np.ctypeslib.as_array(p, (self.n_rows * self.n_cols, self.L))

    def coeffs(self):
        p = C.cast(lib().lo_commit_coeffs(self.h), u64p)
        return np.ctypeslib.as_array(p, (self.n_rows * self.n_per_row, self.L))

    def hashes(self):
        p = C.cast(lib().lo_commit_hashes(self.h), u8p)
        return np.ctypeslib.as_array(p, (self.n_hashes, 32))

    def get_root(self):
        out = np.zeros(32, np.uint8)
        lib().lo_commit_root(self.h, ptr(out))
        return out.tobytes()

    def merkleize(self, n_threads=1):
        lib().lo_merkleize(self.h, n_threads)

    def merkleize_ser(self):
        lib().lo_merkleize_ser(self.h)

    def collapse(self, tensor, n_threads=1):
        tensor = np.ascontiguousarray(tensor, np.uint64)
        poly = np.zeros((self.n_per_row, self.L), np.uint64)
        rc = lib().lo_collapse_columns(self.h, ptr(tensor), tensor.size // self.L, ptr(poly), n_threads)
        if rc:
            raise RuntimeError("oracle collapse failed: %d" % rc)
        return poly

    def open_column(self, col):
        path_len = max(0, (self.n_cols - 1).bit_length())
        cv = np.zeros((self.n_rows, self.L), np.uint64)
        pth = np.zeros((max(1, path_len), 32), np.uint8)
        rc = lib().lo_open_column(self.h, col, ptr(cv), ptr(pth))
        if rc:
            raise RuntimeError("oracle open_column failed: %d" % rc)
        return cv, pth[:path_len]

    def prove(self, outer, enc, tr):
        outer = np.ascontiguousarray(outer, np.uint64)
        pp, plen = C.c_void_p(), C.c_uint64()
        cols = np.zeros(enc.get_n_col_opens(), np.uint64)
        rc = lib().lo_prove(self.h, enc.h, ptr(outer), outer.size // self.L, tr.h, C.byref(pp), C.byref(plen), ptr(cols))
        if rc:
            raise RuntimeError("oracle prove failed: %d" % rc)
        proof = C.string_at(pp, plen.value)
        lib().lo_free(pp)
        return proof, cols

    def __del__(self):
        try:
            lib().lo_commit_free(self.h)
        except Exception:
            pass


def verify(enc, root, outer, inner, proof, tr):
    """returns (rc, eval limbs)."""
    outer = np.ascontiguousarray(outer, np.uint64)
    inner = np.ascontiguousarray(inner, np.uint64)
    out = np.zeros(enc.L, np.uint64)
    pb = np.frombuffer(proof, np.uint8)
    rootb = np.frombuffer(root, np.uint8)
    rc = lib().lo_verify(enc.h, ptr(rootb), ptr(outer), outer.size // enc.L, ptr(inner), inner.size // enc.L,
                         ptr(pb), len(proof), tr.h, ptr(out))
    return rc, out


def hash_column(fid, col):
    col = np.ascontiguousarray(col, np.uint64)
    out = np.zeros(32, np.uint8)
    lib().lo_hash_column(fid, ptr(col), col.size // limbs(fid), ptr(out))
    return out.tobytes()


def leaf_chunk_cvs(fid, comm_local, n_cols, row_base, n_rows_local, n_rows_total, chunk_begin, chunk_end):
    comm_local = np.ascontiguousarray(comm_local, np.uint64)
    out = np.zeros((max(chunk_end - chunk_begin, 0), n_cols, 32), np.uint8)
    if out.size:
        rc = lib().lo_leaf_chunk_cvs(fid, ptr(comm_local), n_cols, row_base, n_rows_local, n_rows_total, chunk_begin, chunk_end, ptr(out))
        if rc:
            raise RuntimeError("lo_leaf_chunk_cvs: %d" % rc)
    return out


def finish_from_cvs(all_cvs, n_cols):
    all_cvs = np.ascontiguousarray(all_cvs, np.uint8)
    n_chunks = all_cvs.shape[0]
    np2 = 1 << max(0, (n_cols - 1).bit_length())
    hashes = np.zeros((2 * np2 - 1, 32), np.uint8)
    lib().lo_finish_from_cvs(ptr(all_cvs), n_chunks, n_cols, ptr(hashes))
    return hashes


def commit_streaming(enc, n_coeffs, rows_of, n_threads=1, chunks_per_step=1):
    """LcCommit::commit (lcpc-2d/src/lib.rs:622-671) WITHOUT holding comm: the rows are encoded a chunk-aligned block at a time
    (lib.rs:648-653), each block is reduced to the BLAKE3 chunk chaining values of every column's leaf message
    (hash_columns, lib.rs:706-745: the digest of 0^32 || repr(col[0]) || ... is a tree over 1 KiB chunks) and dropped; the
    leaf digests and the Merkle tree (lib.rs:747-785) follow from the chaining values.  Host memory: one block of encoded
    rows + 32 bytes per (chunk, column) -- ~2 GB at 2^28 Ft255 coefficients where Commit.commit needs ~40 GB.
    rows_of(r0, r1) returns the coefficients of rows [r0, r1) as an (elements, L) uint64 array (a ragged last row is padded
    here).  Returns the `hashes` array (2 * np2 - 1, 32).  chunks_per_step counts the possible cuts (every chunk; every third for
    Ft191, whose 24-byte elements straddle the others)."""
    fid, L = enc.fid, enc.L
    F = 8 * L
    n_rows, n_per_row, n_cols = enc.get_dims(n_coeffs)
    n_chunks = (32 + F * n_rows + 1023) // 1024
    # a block of rows may end only where a chunk boundary is a row boundary: every chunk when F | 1024, every third for Ft191
    cuts = [c for c in range(n_chunks + 1) if c in (0, n_chunks) or (1024 * c - 32) % F == 0]
    first_row = lambda ch: 0 if ch == 0 else min(n_rows, (ch * 1024 - 32) // F)
    cvs = np.zeros((n_chunks, n_cols, 32), np.uint8)
    for i in range(0, len(cuts) - 1, chunks_per_step):
        cb, ce = cuts[i], cuts[min(i + chunks_per_step, len(cuts) - 1)]
        r0, r1 = first_row(cb), (n_rows if ce >= n_chunks else first_row(ce))
        blk = np.zeros(((r1 - r0) * n_per_row, L), np.uint64)
        got = np.ascontiguousarray(rows_of(r0, r1), np.uint64).reshape(-1, L)
        blk[:got.shape[0]] = got
        comm = np.empty(((r1 - r0) * n_cols, L), np.uint64)
        if lib().lo_encode_rows(enc.h, ptr(blk), r1 - r0, n_threads, ptr(comm)):
            raise RuntimeError("lo_encode_rows failed")
        out = cvs[cb:ce]
        if lib().lo_leaf_chunk_cvs_mt(fid, ptr(comm), n_cols, r0, r1 - r0, n_rows, cb, ce, ptr(out), n_threads):
            raise RuntimeError("lo_leaf_chunk_cvs_mt failed")
        del comm, blk, got
    np2 = 1 << max(0, (n_cols - 1).bit_length())
    hashes = np.zeros((2 * np2 - 1, 32), np.uint8)
    lib().lo_finish_from_cvs_mt(ptr(cvs), n_chunks, n_cols, ptr(hashes), n_threads)
    return hashes
