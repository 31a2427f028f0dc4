"""CPU-side checks of the product library: it loads, exports every symbol include/lcpc_hip.h declares,
fails loudly without a GPU (no CPU fallback), and its host-side dims logic matches the oracle."""
import os
import re

import pytest

import lcpc_amd
from lcpc_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    src = open(os.path.join(ROOT, "include", "lcpc_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(lcpc_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    syms = header_symbols()
    assert len(syms) >= 35
    L = _lib.lib()
    for s in syms:
        assert hasattr(L, s), "missing export: " + s
    assert sorted(_lib.SYMBOLS) == syms, "ctypes table and header disagree"
    assert L.lcpc_abi_version() == _lib.ABI_VERSION == 5
    hdr = open(os.path.join(ROOT, "include", "lcpc_hip.h")).read()
    assert int(re.search(r"#define LCPC_ABI_VERSION (\d+)", hdr).group(1)) == _lib.ABI_VERSION


def test_no_cpu_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(lcpc_amd.LcpcError) as e:
        lcpc_amd.LigeroEncoding.new(lcpc_amd.FT63, 1 << 10)
    assert e.value.code == -18          # LCPC_ERR_NO_DEVICE


def test_product_never_imports_oracle():
    """the product (lcpc_amd/, include/) must not import, link, load or even name the CPU oracle: no file mentions it,
    no Python module imports anything outside the package / stdlib / numpy / torch, and the shared library neither
    needs nor dlopen()s it."""
    import ast
    import subprocess
    allowed_mods = {"ctypes", "os", "sys", "subprocess", "numpy", "torch", "lcpc_amd", ""}     # (sys: the build-time generators under csrc/gen)
    for top in ("lcpc_amd", "include"):
        for dirpath, _, files in os.walk(os.path.join(ROOT, top)):
            for f in files:
                if not f.endswith((".py", ".cpp", ".hip", ".h", ".c")) and f != "Makefile":
                    continue
                txt = open(os.path.join(dirpath, f)).read()
                for word in ("oracle", "pyref", "tests/golden"):
                    assert word not in txt.lower(), "%s mentions %r" % (f, word)
                if f.endswith(".py"):
                    for node in ast.walk(ast.parse(txt)):
                        mods = []
                        if isinstance(node, ast.Import):
                            mods = [a.name for a in node.names]
                        elif isinstance(node, ast.ImportFrom):
                            mods = [node.module or ""] if node.level == 0 else []
                        for m in mods:
                            assert m.split(".")[0] in allowed_mods, "%s imports %s" % (f, m)
    out = subprocess.run(["readelf", "-d", _lib.LIB_PATH], capture_output=True, text=True).stdout
    needed = re.findall(r"NEEDED.*\[(.*?)\]", out)
    assert needed and not any("oracle" in n for n in needed), needed
    blob = open(_lib.LIB_PATH, "rb").read()
    assert b"liblcpc_oracle" not in blob and b"lcpc_oracle" not in blob


def test_static_dims_match_oracle(oracle):
    import ctypes as C
    import random
    rnd = random.Random(2)
    for fid in (0, 3):
        for rho in ((1, 2), (1, 4), (38, 39)):
            for _ in range(60):
                n = rnd.randrange(2, 1 << rnd.randrange(2, 30))
                a, b, c = C.c_uint64(), C.c_uint64(), C.c_uint64()
                assert oracle.lib().lo_ligero_get_dims(fid, n, rho[0], rho[1], C.byref(a), C.byref(b), C.byref(c)) == 0
                assert lcpc_amd.static_get_dims(fid, lcpc_amd.ENC_LIGERO, n, rho=rho) == (a.value, b.value, c.value)
        for code in range(1, 7):
            for _ in range(20):
                n = rnd.randrange(50, 1 << rnd.randrange(7, 30))
                a, b, c = C.c_uint64(), C.c_uint64(), C.c_uint64()
                assert oracle.lib().lo_sdig_get_dims(fid, n, code, C.byref(a), C.byref(b), C.byref(c)) == 0
                assert lcpc_amd.static_get_dims(fid, lcpc_amd.ENC_SDIG, n, code=code) == (a.value, b.value, c.value)


def test_new_ml_dims_match_oracle(oracle):
    """LigeroEncoding::new_ml / SdigEncoding::new_ml shapes (lcpc_static_get_dims_ml) == the oracle's, incl. where
    the reference's assert!s reject the split."""
    import ctypes as C
    for fid in (0, 3):
        for n_vars in range(1, 31):
            a, b, c = C.c_uint64(), C.c_uint64(), C.c_uint64()
            for rho in ((1, 2), (1, 4), (38, 39)):
                rc = oracle.lib().lo_ligero_get_dims_ml(fid, n_vars, rho[0], rho[1], C.byref(a), C.byref(b), C.byref(c))
                if rc == 0:
                    assert lcpc_amd.static_get_dims_ml(fid, lcpc_amd.ENC_LIGERO, n_vars, rho=rho) == (a.value, b.value, c.value)
                else:
                    with pytest.raises(lcpc_amd.LcpcError):
                        lcpc_amd.static_get_dims_ml(fid, lcpc_amd.ENC_LIGERO, n_vars, rho=rho)
            for code in range(1, 7):
                if n_vars < 6:
                    continue
                assert oracle.lib().lo_sdig_get_dims_ml(fid, n_vars, code, C.byref(a), C.byref(b), C.byref(c)) == 0
                assert lcpc_amd.static_get_dims_ml(fid, lcpc_amd.ENC_SDIG, n_vars, code=code) == (a.value, b.value, c.value)


def test_constructor_dims_fixture():
    """the product's dims optimisers against tests/golden/constructor_dims.json (host-only: no device needed)."""
    import json
    import os
    rows = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "constructor_dims.json")))
    for r in rows:
        fid, lg = r["field"], r["log_len"]
        if r["enc"] == "ligero":
            kw = dict(rho=tuple(r["rho"]))
            enc = lcpc_amd.ENC_LIGERO
        else:
            kw = dict(code=r["code"])
            enc = lcpc_amd.ENC_SDIG
        assert list(lcpc_amd.static_get_dims(fid, enc, 1 << lg, **kw)) == r["new"]
        if r["new_ml"] is None:
            with pytest.raises(lcpc_amd.LcpcError):
                lcpc_amd.static_get_dims_ml(fid, enc, lg, **kw)
        else:
            assert list(lcpc_amd.static_get_dims_ml(fid, enc, lg, **kw)) == r["new_ml"]


def test_transcript_matches_oracle(oracle):
    import random
    rnd = random.Random(4)
    t1, t2 = lcpc_amd.Transcript(b"test protocol"), oracle.Transcript(b"test protocol")
    t1.append_message(b"some label", b"some data")
    t2.append_message(b"some label", b"some data")
    assert t1.challenge_bytes(b"challenge", 32).hex() == "d5a21972d0d5fe320c0d263fac7fffb8145aa640af6e9bca177c03c7efcf0615"
    t2.challenge_bytes(b"challenge", 32)
    for _ in range(30):
        m = bytes(rnd.randrange(256) for _ in range(rnd.randrange(0, 400)))
        t1.append_message(b"lbl", m)
        t2.append_message(b"lbl", m)
        k = rnd.randrange(1, 300)
        assert t1.challenge_bytes(b"ch", k) == t2.challenge_bytes(b"ch", k)
    t3 = t1.clone()
    assert t3.challenge_bytes(b"x", 16) == t1.challenge_bytes(b"x", 16)


def test_batched_absorb_matches_oracle(oracle):
    """lcpc_transcript_append_messages (sponge kept in vector registers, AVX-512VL Keccak when the CPU has it) == the
    oracle's append_message loop, across block-boundary alignments, message lengths and counts; the transcripts stay
    interchangeable afterwards."""
    import random
    rnd = random.Random(9)
    for trial in range(60):
        mlen = [8, 16, 24, 32][trial % 4] if trial < 40 else rnd.randrange(1, 150)
        n = rnd.randrange(1, 2500)
        label = bytes(rnd.randrange(256) for _ in range(rnd.randrange(1, 9)))
        data = bytes(rnd.randrange(256) for _ in range(n * mlen))
        pre = bytes(rnd.randrange(256) for _ in range(rnd.randrange(0, 60)))
        t1, t2 = lcpc_amd.Transcript(b"batched"), oracle.Transcript(b"batched")
        t1.append_message(b"pre", pre)
        t2.append_message(b"pre", pre)
        t1.append_messages(label, data, mlen)
        for i in range(n):
            t2.append_message(label, data[i * mlen:(i + 1) * mlen])
        assert t1.challenge_bytes(b"c", 48) == t2.challenge_bytes(b"c", 48), (trial, mlen, n)
        t1.append_message(b"post", pre)
        t2.append_message(b"post", pre)
        assert t1.challenge_bytes(b"d", 16) == t2.challenge_bytes(b"d", 16)


@pytest.mark.parametrize("mix", ["portable", "tern", "xor"])
def test_keccak_mixes_match_oracle(mix):
    """every Keccak-f[1600] implementation in host_crypto.cpp (scalar, and the two generated AVX-512VL instruction mixes of
    which one is picked per CPU vendor) gives the oracle's transcript: run in a fresh process with LCPC_KECCAK forcing it."""
    import subprocess
    import sys
    code = r"""
import os, random, sys
sys.path[:0] = [%r, %r, %r]
import oracle_lib as O
import lcpc_amd
rnd = random.Random(5)
for trial in range(12):
    mlen = [8, 32, 24, 77][trial %% 4]
    n = rnd.randrange(1, 1500)
    data = bytes(rnd.randrange(256) for _ in range(n * mlen))
    t1, t2 = lcpc_amd.Transcript(b"mix"), O.Transcript(b"mix")
    t1.append_message(b"pre", data[:37]); t2.append_message(b"pre", data[:37])
    t1.append_messages(b"lbl", data, mlen)
    for i in range(n):
        t2.append_message(b"lbl", data[i * mlen:(i + 1) * mlen])
    assert t1.challenge_bytes(b"c", 64) == t2.challenge_bytes(b"c", 64), trial
    for i in range(40):                      # the one-message path (keccak_f1600 through run_f)
        t1.append_message(b"x", data[i:i + 50]); t2.append_message(b"x", data[i:i + 50])
    assert t1.challenge_bytes(b"d", 32) == t2.challenge_bytes(b"d", 32)
print("ok")
""" % (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle"))
    env = dict(os.environ, LCPC_KECCAK=mix)
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    if mix != "portable" and "Illegal instruction" in (r.stderr or "") + str(r.returncode):
        pytest.skip("no AVX-512VL on this CPU")
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stderr[-2000:]


def test_shard_node_layout_matches_python():
    """lcpc_shard_nodes (C) == lcpc_amd.distributed.aligned_nodes (Python): both sides of the exchange must agree."""
    import ctypes as C
    from lcpc_amd.distributed import aligned_nodes, chunk_split
    L = _lib.lib()
    for n_chunks in list(range(1, 40)) + [65, 129, 1000]:
        for world in (1, 2, 3, 4, 8):
            for rank, (b, e) in enumerate(chunk_split(n_chunks, world)):
                n = C.c_uint32()
                first = (C.c_uint64 * 64)()
                lg = (C.c_uint32 * 64)()
                assert L.lcpc_shard_nodes(n_chunks, world, rank, C.byref(n), first, lg) == 0
                assert [(first[i], lg[i]) for i in range(n.value)] == aligned_nodes(b, e)
            # the field-aware form: the same for element sizes that divide 1024; for Ft191 (24 bytes) the cuts move down to the
            # chunks that start on a row boundary (chunk = 2 mod 3)
            for field, eb in ((0, 8), (1, 16), (2, 24), (3, 32)):
                sp = chunk_split(n_chunks, world, eb)
                assert sp[0][0] == 0 and sp[-1][1] == n_chunks and all(sp[i][1] == sp[i + 1][0] for i in range(len(sp) - 1))
                if eb != 24:
                    assert sp == chunk_split(n_chunks, world)
                for rank, (b, e) in enumerate(sp):
                    if eb == 24 and 0 < b < n_chunks:
                        assert b % 3 == 2
                    n = C.c_uint32()
                    first = (C.c_uint64 * 64)()
                    lg = (C.c_uint32 * 64)()
                    assert L.lcpc_shard_nodes_field(field, n_chunks, world, rank, C.byref(n), first, lg) == 0
                    assert [(first[i], lg[i]) for i in range(n.value)] == aligned_nodes(b, e)


def test_generated_sources_are_deterministic_and_current():
    """keccak_x25_gen.h, field_r29_gen.h, field_ln_gen.h and field_wmul_gen.h are build products of lcpc_amd/csrc/Makefile (git-ignored, 13 000 lines):
    each generator is a pure function of its own text -- two runs give the same bytes -- and the header the library was built
    from is what the generator prints now (a stale header would mean the .so does not match the tree)."""
    import subprocess
    import sys
    csrc = os.path.join(ROOT, "lcpc_amd", "csrc")
    tracked = subprocess.run(["git", "-C", ROOT, "ls-files", "lcpc_amd/csrc"], capture_output=True, text=True).stdout
    for gen, hdr in (("gen_keccak_x25.py", "keccak_x25_gen.h"), ("gen_r29_asm.py", "field_r29_gen.h"), ("gen_ln_asm.py", "field_ln_gen.h"),
                     ("gen_wmul_asm.py", "field_wmul_gen.h")):
        a = subprocess.run([sys.executable, os.path.join(csrc, "gen", gen)], capture_output=True, check=True).stdout
        b = subprocess.run([sys.executable, os.path.join(csrc, "gen", gen)], capture_output=True, check=True, env=dict(os.environ, PYTHONHASHSEED="12345")).stdout
        assert a == b and len(a) > 1000, gen
        assert open(os.path.join(csrc, hdr), "rb").read() == a, "%s is stale: rebuild (make -C lcpc_amd/csrc)" % hdr
        assert hdr not in tracked, "%s must not be tracked" % hdr


def test_product_library_has_no_test_hooks():
    """the allocation-failure hooks of ctx.cpp (LCPC_TEST_FAIL) exist only in lib/liblcpc_hip_testhooks.so, the second build the
    Makefile makes with -DLCPC_TEST_HOOKS for tests/common.py run_with_test_hooks; the product neither reads the variable nor
    carries the branches.  Both export the same ABI."""
    import subprocess
    lib_dir = os.path.dirname(_lib.LIB_PATH)
    hooks = os.path.join(lib_dir, "liblcpc_hip_testhooks.so")
    assert os.path.exists(hooks)
    assert b"LCPC_TEST_FAIL" not in open(_lib.LIB_PATH, "rb").read()
    assert b"LCPC_TEST_FAIL" in open(hooks, "rb").read()
    syms = lambda p: sorted(l.split()[-1] for l in subprocess.check_output(["nm", "-D", "--defined-only", p], text=True).splitlines() if " T lcpc_" in l)
    assert syms(hooks) == syms(_lib.LIB_PATH)
