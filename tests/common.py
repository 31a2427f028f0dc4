"""helpers shared by the CPU (oracle) and GPU (HIP) test files."""
import hashlib
import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_golden(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


def hex_to_limbs(h, L):
    v = int(h, 16)
    return np.array([(v >> (64 * k)) & 0xFFFFFFFFFFFFFFFF for k in range(L)], np.uint64)


def golden_coeffs(oracle, case):
    """the deterministic inputs of tests/golden/make_golden.py, as (n, L) Montgomery limbs."""
    fid, n = case["field"], case["n_coeffs"]
    L = oracle.limbs(fid)
    if case["coeffs"] == "iota":
        v = np.arange(1, n + 1, dtype=np.uint64)
        out = np.zeros((n, L), np.uint64)
        oracle.lib().lo_f_from_u64(fid, oracle.ptr(v), oracle.ptr(out), n)
        return out
    return oracle.random_elems(fid, n, case["seed"])


def powers(oracle, fid, x_int, n, start_exp_step=1):
    """[x^(k*step)] for k < n as Montgomery limbs (python ints -> oracle conversion)."""
    import pyref as P
    F = P.FIELDS[fid]
    base = pow(x_int, start_exp_step, F.p)
    vals, cur = [], 1
    for _ in range(n):
        vals.append(cur)
        cur = cur * base % F.p
    return oracle.to_mont(fid, vals)


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def mk_transcript(T, root, n_col_opens):
    # lcpc-ligero-pc/src/tests.rs:243-245
    tr = T(b"test transcript")
    tr.append_message(b"polycommit", bytes(root))
    tr.append_message(b"ncols", int(n_col_opens).to_bytes(8, "big"))
    return tr


def commit_bincode(oc):
    """bincode 1.3 of WrappedLcCommit (lcpc-2d/src/lib.rs:186-197) built from a commitment's fields (oracle or HIP object with
    comm() / coeffs() / hashes() / n_rows / n_cols / n_per_row): Vec<F> = u64 len + raw Montgomery limbs; usize = u64;
    Vec<WrappedOutput> = u64 len + (u64 32 + 32 bytes) each."""
    import struct
    comm, coeffs, hashes = oc.comm(), oc.coeffs(), oc.hashes()
    out = [struct.pack("<Q", comm.shape[0]), comm.tobytes(), struct.pack("<Q", coeffs.shape[0]), coeffs.tobytes(),
           struct.pack("<QQQ", oc.n_rows, oc.n_cols, oc.n_per_row), struct.pack("<Q", hashes.shape[0])]
    for h in hashes:
        out.append(struct.pack("<Q", 32) + bytes(h))
    return b"".join(out)



def run_with_test_hooks(body, env=None, timeout=600):
    """run `body` (python source) in a child process whose lcpc_amd loads lib/liblcpc_hip_testhooks.so -- the library built with
    -DLCPC_TEST_HOOKS (lcpc_amd/csrc/Makefile), the only build that reads LCPC_TEST_FAIL.  The product library carries no such branch."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hooks = os.path.join(root, "lcpc_amd", "lib", "liblcpc_hip_testhooks.so")
    assert os.path.exists(hooks), "lcpc_amd/csrc/Makefile builds it beside the product"
    pre = ("import os, sys\nsys.path[:0] = [%r, %r, %r]\nimport lcpc_amd._lib as _L\n_L.LIB_PATH = %r\n"
           "import numpy as np\nimport oracle_lib as O\nfrom lcpc_amd import LcCommit, LigeroEncoding\n"
           % (root, os.path.join(root, "tests"), os.path.join(root, "oracle"), hooks))
    e = dict(os.environ)
    e.update(env or {})
    r = subprocess.run([sys.executable, "-c", pre + body], capture_output=True, text=True, env=e, timeout=timeout)
    assert r.returncode == 0, r.stderr[-3000:]
    return r.stdout
