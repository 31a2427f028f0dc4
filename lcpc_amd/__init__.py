"""lcpc_amd -- host-side mirror of conroi/lcpc's LcEncoding / LcCommit / prove / verify API over the
MI355X-native C ABI (include/lcpc_hip.h).  The reference's toolchain (Rust) is absent from this image, so
the host side above the C ABI is Python + ctypes with the reference's names, argument meaning and error
behaviour; INTEGRATION.md shows the Rust `extern "C"` binding a maintainer would add instead.

Field elements are numpy uint64 arrays of shape (n, L): ff_derive's Montgomery limbs, exactly what a Rust
`&[FtN]` is in memory.  (/root/reference citations: see include/lcpc_hip.h.)"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import LcpcParams, LcpcTimings

FT63, FT127, FT191, FT255 = 0, 1, 2, 3
FIELD_LIMBS = {FT63: 1, FT127: 2, FT191: 3, FT255: 4}
ENC_LIGERO, ENC_SDIG = 0, 1


class LcpcError(RuntimeError):
    """a negative lcpc_status; .code mirrors ProverError / VerifierError (lcpc-2d/src/lib.rs:111-166)."""

    def __init__(self, code, detail=""):
        self.code = code
        msg = _lib.lib().lcpc_strerror(code).decode()
        super().__init__("%s (%d)%s" % (msg, code, (": " + detail) if detail else ""))


ERR_TOO_BIG, ERR_ENCODE, ERR_COMMIT, ERR_COLUMN_NUMBER, ERR_OUTER_TENSOR, ERR_DIMS, ERR_ARG, ERR_STATE = -1, -2, -3, -4, -5, -6, -7, -8
VERR_NUM_COL_OPENS, VERR_COLUMN_PATH, VERR_COLUMN_EVAL, VERR_COLUMN_DEGREE = -32, -33, -34, -35
VERR_OUTER_TENSOR, VERR_INNER_TENSOR, VERR_ENCODING_DIMS, VERR_ENCODE, VERR_MALFORMED = -36, -37, -38, -39, -40


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def _elems(a, L):
    a = np.ascontiguousarray(a, dtype=np.uint64)
    if a.size % L:
        raise ValueError("element array size is not a multiple of L")
    return a


class Transcript:
    """merlin::Transcript (the `tr: &mut Transcript` of prove/verify)."""

    def __init__(self, label=b"", _h=None):
        self._h = _h if _h is not None else _lib.lib().lcpc_transcript_new(label, len(label))

    def append_message(self, label, message):
        _lib.lib().lcpc_transcript_append_message(self._h, label, len(label), bytes(message), len(message))

    def append_messages(self, label, messages, mlen):
        """append_message(label, m) for every mlen-byte slice m of `messages`."""
        messages = bytes(messages)
        _lib.lib().lcpc_transcript_append_messages(self._h, label, len(label), messages, mlen, len(messages) // mlen)

    def challenge_bytes(self, label, n):
        out = C.create_string_buffer(n)
        _lib.lib().lcpc_transcript_challenge_bytes(self._h, label, len(label), out, n)
        return out.raw

    def clone(self):
        return Transcript(_h=_lib.lib().lcpc_transcript_clone(self._h))

    def __del__(self):
        try:
            _lib.lib().lcpc_transcript_free(self._h)
        except Exception:
            pass


class _Encoding:
    """common part of the two LcEncoding implementors; owns one lcpc_ctx (encoder tables on the GPU)."""

    def __init__(self, params):
        self.params = params
        h = C.c_void_p()
        rc = _lib.lib().lcpc_ctx_create(C.byref(params), C.byref(h))
        if rc:
            raise LcpcError(rc)
        self._h = h
        self.field = params.field
        self.L = FIELD_LIMBS[params.field]
        a, b, c = C.c_uint64(), C.c_uint64(), C.c_uint64()
        _lib.lib().lcpc_get_dims(self._h, 1, C.byref(a), C.byref(b), C.byref(c))
        self.n_per_row, self.n_cols = b.value, c.value

    def _check(self, rc):
        if rc:
            raise LcpcError(rc, _lib.lib().lcpc_last_error(self._h).decode())

    # LcEncoding trait, lcpc-2d/src/lib.rs:74-104
    def get_dims(self, length):
        a, b, c = C.c_uint64(), C.c_uint64(), C.c_uint64()
        self._check(_lib.lib().lcpc_get_dims(self._h, length, C.byref(a), C.byref(b), C.byref(c)))
        return a.value, b.value, c.value

    def dims_ok(self, n_per_row, n_cols):
        return bool(_lib.lib().lcpc_dims_ok(self._h, n_per_row, n_cols))

    def get_n_col_opens(self):
        return _lib.lib().lcpc_get_n_col_opens(self._h)

    def get_n_degree_tests(self):
        return _lib.lib().lcpc_get_n_degree_tests(self._h)

    def encode(self, rows):
        """in place on a copy: rows of n_cols elements, message in the first n_per_row, rest zero."""
        rows = _elems(rows, self.L).copy()
        n = rows.size // (self.L * self.n_cols)
        if n * self.L * self.n_cols != rows.size:
            raise LcpcError(ERR_ENCODE, "row length != n_cols")
        self._check(_lib.lib().lcpc_encode_rows(self._h, _ptr(rows), n))
        return rows

    def random_coeffs_device(self, n, seed=0, stream_id=0, out_ptr=None, stream=0):
        """lcpc_test_fields::random_coeffs (lcpc-test-fields/src/lib.rs:75-97) with a fixed generator, on the device: n elements by
        Field::random from ChaCha20Rng::from_seed([seed; 32]), stream `stream_id` (lcpc_random_coeffs_device) -- element for
        element the vector the same rule gives on a host.  Returns a torch int64 tensor [n, L] on the encoder's device, or
        fills `out_ptr` (a device address with room for n * L u64) and returns None."""
        key = (C.c_uint8 * 32)(*([seed & 0xFF] * 32))
        t = None
        if out_ptr is None:
            import torch
            t = torch.empty((n, self.L), dtype=torch.int64, device=torch.device("cuda", self.params.device))
            out_ptr = t.data_ptr()
        self._check(_lib.lib().lcpc_random_coeffs_device(self._h, key, stream_id, n, C.c_void_p(out_ptr), C.c_void_p(stream)))
        return t

    def __del__(self):
        try:
            _lib.lib().lcpc_ctx_destroy(self._h)      # commitments made with it keep the tables alive (refcount)
        except Exception:
            pass


def _params(field, encoding, device, **kw):
    p = LcpcParams()
    p.field, p.encoding, p.hash, p.device = field, encoding, 0, device
    p.rho_num, p.rho_den = kw.get("rho", (1, 2))
    p.sdig_code, p.seed = kw.get("code", 3), kw.get("seed", 0)
    p.n_coeffs, p.n_per_row, p.n_cols = kw.get("n_coeffs", 0), kw.get("n_per_row", 0), kw.get("n_cols", 0)
    p.shard_rank, p.shard_count = kw.get("shard", (0, 1))
    return p


def static_get_dims(field, encoding, length, rho=(1, 2), code=3):
    """LigeroEncodingRho::_get_dims (ligero lib.rs:70-112) / SdigEncodingS::new's shape (brakedown lib.rs:69-110)."""
    p = _params(field, encoding, 0, n_coeffs=length, rho=rho, code=code)
    a, b, c = C.c_uint64(), C.c_uint64(), C.c_uint64()
    rc = _lib.lib().lcpc_static_get_dims(C.byref(p), C.byref(a), C.byref(b), C.byref(c))
    if rc:
        raise LcpcError(rc)
    return a.value, b.value, c.value


def static_get_dims_ml(field, encoding, n_vars, rho=(1, 2), code=3):
    """the dims new_ml(n_vars) picks (ligero lib.rs:128-135, brakedown lib.rs:114-123); LcpcError(ERR_DIMS) where the
    reference's assert!s fire."""
    p = _params(field, encoding, 0, n_coeffs=1, rho=rho, code=code)
    a, b, c = C.c_uint64(), C.c_uint64(), C.c_uint64()
    rc = _lib.lib().lcpc_static_get_dims_ml(C.byref(p), n_vars, C.byref(a), C.byref(b), C.byref(c))
    if rc:
        raise LcpcError(rc)
    return a.value, b.value, c.value


class LigeroEncoding(_Encoding):
    """LigeroEncodingRho<Ft, Rn, Rd> (lcpc-ligero-pc/src/lib.rs:31-186); default rate 1/2 (lib.rs:189)."""

    def __init__(self, field, length=None, rho=(1, 2), device=0, shard=(0, 1), _dims=None):
        kw = dict(rho=rho, shard=shard)
        if _dims is not None:
            kw.update(n_per_row=_dims[0], n_cols=_dims[1])
        else:
            kw.update(n_coeffs=length)
        super().__init__(_params(field, ENC_LIGERO, device, **kw))

    @classmethod
    def new(cls, field, length, rho=(1, 2), device=0, shard=(0, 1)):
        return cls(field, length, rho, device, shard)

    @classmethod
    def new_ml(cls, field, n_vars, rho=(1, 2), device=0):
        """LigeroEncodingRho::new_ml (lib.rs:128-135)."""
        _, n_per_row, n_cols = static_get_dims_ml(field, ENC_LIGERO, n_vars, rho)
        return cls.new_from_dims(field, n_per_row, n_cols, rho, device)

    @classmethod
    def new_from_dims(cls, field, n_per_row, n_cols, rho=(1, 2), device=0, shard=(0, 1)):
        return cls(field, None, rho, device, shard, _dims=(n_per_row, n_cols))


class SdigEncoding(_Encoding):
    """SdigEncodingS<Ft, S> (lcpc-brakedown-pc/src/lib.rs:41-176); default code SdigCode3 (lib.rs:19)."""

    def __init__(self, field, length=None, seed=0, code=3, device=0, shard=(0, 1), _dims=None):
        kw = dict(seed=seed, code=code, shard=shard)
        if _dims is not None:
            kw.update(n_per_row=_dims[0], n_cols=_dims[1])
        else:
            kw.update(n_coeffs=length)
        super().__init__(_params(field, ENC_SDIG, device, **kw))

    @classmethod
    def new(cls, field, length, seed, code=3, device=0):
        return cls(field, length, seed, code, device)

    @classmethod
    def new_ml(cls, field, n_vars, seed, code=3, device=0):
        """SdigEncodingS::new_ml (lib.rs:114-123)."""
        _, n_per_row, n_cols = static_get_dims_ml(field, ENC_SDIG, n_vars, code=code)
        return cls.new_from_dims(field, n_per_row, n_cols, seed, code, device)

    @classmethod
    def new_from_dims(cls, field, n_per_row, n_cols, seed, code=3, device=0):
        return cls(field, None, seed, code, device, _dims=(n_per_row, n_cols))


BORROW_COEFFS = 1      # LCPC_COMMIT_BORROW_COEFFS
ASYNC_TAIL = 2         # LCPC_COMMIT_ASYNC_TAIL (lcpc_commit_sharded_device)


class LcCommit:
    """LcCommit<D, E> (lcpc-2d/src/lib.rs:172-184, 270-312) with D = BLAKE3: one lcpc_commit_t -- comm / coeffs / hashes
    of ONE commitment, resident in HBM.  Any number of them may be live under one encoding object (lib.rs:299-311)."""

    def __init__(self, enc):
        """an empty commitment bound to `enc` (lcpc_commit_create); the commit constructors below fill it."""
        self.enc = enc
        h = C.c_void_p()
        rc = _lib.lib().lcpc_commit_create(enc._h, C.byref(h))
        if rc:
            raise LcpcError(rc)
        self._h = h
        self.n_rows = self.n_per_row = self.n_cols = self.n_hashes = 0

    def _check(self, rc):
        if rc:
            raise LcpcError(rc, _lib.lib().lcpc_commit_last_error(self._h).decode())

    def _refresh(self):
        a, b, c, d = (C.c_uint64() for _ in range(4))
        self._check(_lib.lib().lcpc_commit_dims(self._h, C.byref(a), C.byref(b), C.byref(c), C.byref(d)))
        self.n_rows, self.n_per_row, self.n_cols, self.n_hashes = a.value, b.value, c.value, d.value
        return self

    @classmethod
    def commit(cls, coeffs, enc, into=None):
        """LcCommit::commit(&coeffs, &enc) (lib.rs:299-301) from host memory.  `into`: refill that object (buffers reused)."""
        coeffs = _elems(coeffs, enc.L)
        cm = into if into is not None else cls(enc)
        cm._check(_lib.lib().lcpc_commit(cm._h, _ptr(coeffs), coeffs.size // enc.L, None))
        return cm._refresh()

    @classmethod
    def commit_device(cls, coeffs_ptr, n_coeffs, enc, stream=0, sync=True, borrow=False, into=None):
        """same, coefficients already in HBM (`coeffs_ptr` = device address, e.g. torch tensor .data_ptr()).
        borrow: LCPC_COMMIT_BORROW_COEFFS -- the commitment keeps reading the caller's buffer instead of copying it."""
        cm = into if into is not None else cls(enc)
        root = (C.c_uint8 * 32)() if sync else None
        cm._check(_lib.lib().lcpc_commit_device(cm._h, C.c_void_p(coeffs_ptr), n_coeffs, C.c_void_p(stream),
                                                BORROW_COEFFS if borrow else 0, root))
        return cm._refresh()

    @classmethod
    def from_parts(cls, enc, comm, coeffs, n_rows, into=None):
        """test hook = lcpc-2d/src/tests.rs:435-466 random_comm + merkleize."""
        comm = _elems(comm, enc.L)
        cp = _ptr(_elems(coeffs, enc.L)) if coeffs is not None else None
        cm = into if into is not None else cls(enc)
        cm._check(_lib.lib().lcpc_commit_from_parts(cm._h, _ptr(comm), cp, n_rows, None))
        return cm._refresh()

    # ---- serde of LcCommit (lcpc-2d/src/lib.rs:186-268), bincode 1.3, streamed ----
    def bincode_size(self):
        return int(_lib.lib().lcpc_commit_bincode_size(self._h))

    def to_bincode(self, fileobj):
        """bincode::serialize_into(fileobj, &commit): the bytes the reference's `Deserialize for LcCommit` takes."""
        err = []

        def wr(_user, data, n):
            try:
                fileobj.write(C.string_at(data, n))
                return 0
            except Exception as e:      # never unwind through the C frame
                err.append(e)
                return 1

        rc = _lib.lib().lcpc_commit_bincode_write(self._h, _lib.WRITE_FN(wr), None)
        if err:
            raise err[0]
        self._check(rc)

    @classmethod
    def from_bincode(cls, enc, fileobj):
        """bincode::deserialize_from(fileobj) of a commitment the reference (or to_bincode) serialised, into HBM."""
        cm = cls(enc)
        err = []

        def rd(_user, data, n):
            try:
                b = fileobj.read(n)
                if len(b) != n:
                    return 1
                C.memmove(data, b, n)
                return 0
            except Exception as e:
                err.append(e)
                return 1

        rc = _lib.lib().lcpc_commit_from_bincode(cm._h, _lib.WRITE_FN(rd), None, None)
        if err:
            raise err[0]
        cm._check(rc)
        return cm._refresh()

    def get_root(self):
        out = (C.c_uint8 * 32)()
        self._check(_lib.lib().lcpc_get_root(self._h, out))
        return bytes(out)

    def get_n_rows(self):
        return self.n_rows

    def get_n_cols(self):
        return self.n_cols

    def get_n_per_row(self):
        return self.n_per_row

    def hashes(self):
        out = np.zeros((self.n_hashes, 32), np.uint8)
        self._check(_lib.lib().lcpc_get_hashes(self._h, _ptr(out)))
        return out

    def comm(self, row0=0, n_rows=None):
        n_rows = self.n_rows - row0 if n_rows is None else n_rows
        out = np.zeros((n_rows * self.n_cols, self.enc.L), np.uint64)
        self._check(_lib.lib().lcpc_get_comm(self._h, row0, n_rows, _ptr(out)))
        return out

    def coeffs(self, row0=0, n_rows=None):
        n_rows = self.n_rows - row0 if n_rows is None else n_rows
        out = np.zeros((n_rows * self.n_per_row, self.enc.L), np.uint64)
        self._check(_lib.lib().lcpc_get_coeffs(self._h, row0, n_rows, _ptr(out)))
        return out

    def eval_outer(self, tensors):
        """collapse_columns (lib.rs:1095-1123) for one (n_rows, L) or several (k, n_rows, L) tensors."""
        t = _elems(tensors, self.enc.L)
        k = t.size // (self.n_rows * self.enc.L) if self.n_rows else 0
        if k == 0 or k * self.n_rows * self.enc.L != t.size:
            raise LcpcError(ERR_OUTER_TENSOR)
        out = np.zeros((k, self.n_per_row, self.enc.L), np.uint64)
        self._check(_lib.lib().lcpc_collapse(self._h, _ptr(t), k, _ptr(out)))
        return out[0] if k == 1 and np.ndim(tensors) <= 2 else out

    def open_columns(self, cols):
        cols = np.ascontiguousarray(cols, np.uint64)
        n = cols.size
        path_len = max(0, (self.n_cols - 1).bit_length())
        vals = np.zeros((n, self.n_rows, self.enc.L), np.uint64)
        paths = np.zeros((n, max(path_len, 1), 32), np.uint8)
        self._check(_lib.lib().lcpc_open_columns(self._h, _ptr(cols), n, _ptr(vals), _ptr(paths)))
        return vals, paths[:, :path_len]

    def open_column(self, col):
        v, p = self.open_columns([col])
        return v[0], p[0]

    def prove(self, outer_tensor, enc, tr):
        """LcCommit::prove (lib.rs:304-311): returns an LcEvalProof (bincode bytes + parsed header)."""
        if enc is not self.enc:
            raise LcpcError(ERR_COMMIT)
        t = _elems(outer_tensor, enc.L)
        pp, plen = C.c_void_p(), C.c_uint64()
        cols = np.zeros(enc.get_n_col_opens(), np.uint64)
        self._check(_lib.lib().lcpc_prove(self._h, _ptr(t), t.size // enc.L, tr._h, C.byref(pp), C.byref(plen), _ptr(cols)))
        return LcEvalProof(_OwnedBuffer(pp, plen.value), enc.L, cols)

    def set_timing(self, on=True):
        self._check(_lib.lib().lcpc_set_timing(self._h, 1 if on else 0))

    def timings(self):
        t = LcpcTimings()
        self._check(_lib.lib().lcpc_get_timings(self._h, C.byref(t)))
        return t

    def __del__(self):
        try:
            _lib.lib().lcpc_commit_destroy(self._h)
        except Exception:
            pass


class _OwnedBuffer:
    """a malloc'ed buffer handed out by the library (lcpc_prove): viewed in place, released with lcpc_free"""

    def __init__(self, ptr, n):
        self.ptr, self.n = ptr, n
        self.view = (C.c_uint8 * n).from_address(ptr.value)

    def __del__(self):
        try:
            _lib.lib().lcpc_free(self.ptr)
        except Exception:
            pass


class LcEvalProof:
    """LcEvalProof<D, E> (lib.rs:490-500) held in the reference's bincode wire layout (lib.rs:550-609)."""

    def __init__(self, data, L, cols_opened=None):
        self._own = data if isinstance(data, _OwnedBuffer) else None
        self._bytes = None if self._own else bytes(data)
        self.L, self.cols_opened = L, cols_opened
        buf = self._own.view if self._own else self._bytes
        if len(buf) < 16:         # bincode::deserialize fails on a buffer that ends inside the header (io::UnexpectedEof)
            raise LcpcError(VERR_MALFORMED, "proof shorter than its header")
        hdr = np.frombuffer(buf, np.uint64, 2)
        self.n_cols, self._n_per_row = int(hdr[0]), int(hdr[1])

    @property
    def data(self):
        if self._bytes is None:
            self._bytes = bytes(self._own.view)
        return self._bytes

    def get_n_cols(self):
        return self.n_cols

    def get_n_per_row(self):
        return self._n_per_row

    def to_bytes(self):           # bincode::serialize(&pf)
        return self.data

    @classmethod
    def from_bytes(cls, data, L):  # bincode::deserialize
        return cls(data, L)

    def verify(self, root, outer_tensor, inner_tensor, enc, tr):
        """LcEvalProof::verify (lib.rs:518-527); returns the evaluation (L limbs) or raises LcpcError(VERR_*)."""
        o, i = _elems(outer_tensor, enc.L), _elems(inner_tensor, enc.L)
        out = np.zeros(enc.L, np.uint64)
        buf = np.frombuffer(self._own.view if self._own else self._bytes, np.uint8)
        rootb = np.frombuffer(bytes(root), np.uint8)
        rc = _lib.lib().lcpc_verify(enc._h, _ptr(rootb), _ptr(o), o.size // enc.L, _ptr(i), i.size // enc.L,
                                    _ptr(buf), buf.size, tr._h, _ptr(out))
        if rc:
            raise LcpcError(rc)
        return out


def root_bincode(root):
    out = (C.c_uint8 * 40)()
    _lib.lib().lcpc_root_bincode(bytes(root), out)
    return bytes(out)
