"""The Rust side of the boundary (bindings/rust/) against include/lcpc_hip.h, without a Rust toolchain (the image has none).

lcpc-hip-sys/src/lib.rs declares the C ABI by hand; a wrong argument there is undefined behaviour the day the crate is built.
This test parses the `extern "C"` block, the #[repr(C)] structs, the callback typedefs and the constants of that file and the
declarations of the header, and requires them to agree item by item: the same set of symbols (== what liblcpc_hip.so exports),
the same number of arguments, each argument and return type the Rust spelling of the C type, struct fields in the same order with
the same types, struct sizes equal to gcc's sizeof, every enum constant with the same value.  It also checks that the high-level
crate (lcpc-hip) only calls symbols the sys crate declares, and implements LcEncoding for both encodings.  CPU only."""
import os
import re
import subprocess
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HDR = os.path.join(ROOT, "include", "lcpc_hip.h")
SYS = os.path.join(ROOT, "bindings", "rust", "lcpc-hip-sys", "src", "lib.rs")
HI = os.path.join(ROOT, "bindings", "rust", "lcpc-hip", "src", "lib.rs")

BASE = {"uint64_t": "u64", "uint32_t": "u32", "uint8_t": "u8", "int32_t": "i32", "int": "c_int", "size_t": "usize", "void": "c_void",
        "char": "c_char", "float": "f32", "lcpc_ctx": "lcpc_ctx", "lcpc_commit_t": "lcpc_commit_t", "lcpc_transcript": "lcpc_transcript",
        "lcpc_params": "lcpc_params", "lcpc_timings": "lcpc_timings", "lcpc_write_fn": "lcpc_write_fn", "lcpc_read_fn": "lcpc_read_fn",
        "lcpc_allgather_fn": "lcpc_allgather_fn"}


def strip_c(src):
    return re.sub(r"/\*.*?\*/", " ", src, flags=re.S)


def strip_rs(src):
    src = re.sub(r"//[^\n]*", "", src)
    return src


def c_type_to_rust(decl):
    """`const uint64_t *coeffs_host`, `uint8_t id[128]`, `void *stream`, `uint8_t **proof`, `int enable` -> Rust type"""
    decl = decl.strip()
    arr = re.search(r"\[[0-9]*\]\s*$", decl)
    if arr:
        decl = decl[:arr.start()].strip()
    const = bool(re.search(r"\bconst\b", decl))
    decl = re.sub(r"\bconst\b", "", decl)
    stars = decl.count("*")
    words = decl.replace("*", " ").split()
    base = words[0]
    if base not in BASE:
        raise AssertionError("unmapped C type in %r" % decl)
    r = BASE[base]
    n_ptr = stars + (1 if arr else 0)
    for i in range(n_ptr):
        # the innermost level carries the const of the pointee; outer levels of `T **` are out-parameters
        r = ("*const " if (const and i == 0) else "*mut ") + r
    return r


def split_args(s):
    out, depth, cur = [], 0, ""
    for ch in s:
        if ch in "([<":
            depth += 1
        elif ch in ")]>":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur)
            cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur)
    return [a.strip() for a in out]


def header_functions():
    src = strip_c(open(HDR).read())
    src = re.sub(r"typedef[^;{]*\([^;]*;", " ", src)         # function-pointer typedefs are checked separately
    fns = {}
    for m in re.finditer(r"([A-Za-z_][\w\s\*]*?)\b(lcpc_[a-z0-9_]+)\s*\(([^;{}]*?)\)\s*;", src):
        ret, name, args = m.group(1).strip(), m.group(2), m.group(3).strip()
        a = [] if args in ("", "void") else [c_type_to_rust(x) for x in split_args(args)]
        r = None if ret == "void" else c_type_to_rust(ret + " x")
        fns[name] = (a, r)
    return fns


def rust_functions():
    src = strip_rs(open(SYS).read())
    blk = re.search(r'extern\s+"C"\s*\{(.*?)\n\}', src, flags=re.S).group(1)
    fns = {}
    for m in re.finditer(r"pub fn (lcpc_[a-z0-9_]+)\s*\((.*?)\)\s*(?:->\s*([^;]+?))?\s*;", blk, flags=re.S):
        name, args, ret = m.group(1), m.group(2), m.group(3)
        a = []
        for x in split_args(args):
            if not x:
                continue
            _, ty = x.split(":", 1)
            a.append(" ".join(ty.split()))
        fns[name] = (a, " ".join(ret.split()) if ret else None)
    return fns


def test_extern_block_equals_header():
    c, r = header_functions(), rust_functions()
    assert len(c) == 58, sorted(c)
    assert sorted(c) == sorted(r), "symbols differ: only in header %s, only in lib.rs %s" % (sorted(set(c) - set(r)), sorted(set(r) - set(c)))
    for name in sorted(c):
        ca, cr = c[name]
        ra, rr = r[name]
        assert len(ca) == len(ra), "%s: %d arguments in the header, %d in lib.rs" % (name, len(ca), len(ra))
        for i, (x, y) in enumerate(zip(ca, ra)):
            assert x == y, "%s, argument %d: header says %s, lib.rs says %s" % (name, i, x, y)
        assert cr == rr, "%s: return type %s vs %s" % (name, cr, rr)


def test_extern_block_equals_library_exports():
    from lcpc_amd import _lib
    out = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = sorted(l.split()[2] for l in out.splitlines() if " T lcpc_" in l)
    assert exported == sorted(rust_functions())


def c_struct_fields(name):
    src = strip_c(open(HDR).read())
    body = re.search(r"typedef struct\s*\{([^}]*)\}\s*%s\s*;" % name, src, flags=re.S).group(1)
    fields = []
    for stmt in body.split(";"):
        stmt = stmt.strip()
        if not stmt:
            continue
        ty, names = stmt.split(None, 1)
        for n in names.split(","):
            fields.append((n.strip(), BASE[ty]))
    return fields


def rust_struct_fields(name):
    src = strip_rs(open(SYS).read())
    m = re.search(r"#\[repr\(C\)\]\s*(?:#\[derive\([^)]*\)\]\s*)?pub struct %s\s*\{(.*?)\}" % name, src, flags=re.S)
    assert m, name
    return [(a, b) for a, b in re.findall(r"pub (\w+)\s*:\s*(\w+)\s*,", m.group(1))]


def test_repr_c_structs_equal_header():
    for s in ("lcpc_params", "lcpc_timings"):
        assert c_struct_fields(s) == rust_struct_fields(s), s
    # the sizes the crate's own unit test asserts are gcc's
    prog = '#include <stdio.h>\n#include "lcpc_hip.h"\nint main(void){printf("%zu %zu\\n", sizeof(lcpc_params), sizeof(lcpc_timings));return 0;}\n'
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "s.c"), "w").write(prog)
        subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), os.path.join(d, "s.c"), "-o", os.path.join(d, "s")], check=True)
        sizes = subprocess.run([os.path.join(d, "s")], capture_output=True, text=True, check=True).stdout.split()
    rs = open(SYS).read()
    assert "size_of::<lcpc_params>(), %s" % sizes[0] in rs and "size_of::<lcpc_timings>(), %s" % sizes[1] in rs, sizes


def test_callback_typedefs_equal_header():
    c = strip_c(open(HDR).read())
    r = strip_rs(open(SYS).read())
    for name in ("lcpc_write_fn", "lcpc_read_fn", "lcpc_allgather_fn"):
        cm = re.search(r"typedef\s+int\s*\(\*%s\)\s*\(([^)]*)\)\s*;" % name, c)
        rm = re.search(r"pub type %s\s*=\s*Option<unsafe extern \"C\" fn\((.*?)\)\s*->\s*c_int>;" % name, r, flags=re.S)
        assert cm and rm, name
        ca = [c_type_to_rust(x) for x in split_args(cm.group(1))]
        ra = [" ".join(x.split(":", 1)[1].split()) for x in split_args(rm.group(1))]
        assert ca == ra, (name, ca, ra)


def test_constants_equal_header():
    c = strip_c(open(HDR).read())
    consts = {}
    for body in re.findall(r"enum\s*\{([^}]*)\}", c) + re.findall(r"typedef enum\s*\{([^}]*)\}", c):
        for item in body.split(","):
            if "=" in item:
                k, v = item.split("=")
                consts[k.strip()] = int(v.strip())
    consts["LCPC_ABI_VERSION"] = int(re.search(r"#define LCPC_ABI_VERSION (\d+)", c).group(1))
    assert len(consts) >= 30
    r = strip_rs(open(SYS).read())
    rc = {k: int(v) for k, v in re.findall(r"pub const (LCPC_\w+)\s*:\s*\w+\s*=\s*(-?\d+)\s*;", r)}
    assert rc == consts, (sorted(set(consts) - set(rc)), sorted(set(rc) - set(consts)), {k: (consts[k], rc[k]) for k in consts if k in rc and consts[k] != rc[k]})
    ver = re.search(r'version = "0\.(\d+)\.', open(os.path.join(os.path.dirname(os.path.dirname(SYS)), "Cargo.toml")).read())
    assert int(ver.group(1)) == consts["LCPC_ABI_VERSION"]


def test_high_level_crate_uses_declared_symbols_only():
    hi = strip_rs(open(HI).read())
    used = set(re.findall(r"sys::(lcpc_[a-z0-9_]+)\b", hi)) - {"lcpc_ctx", "lcpc_commit_t", "lcpc_transcript", "lcpc_params", "lcpc_timings"}
    declared = set(rust_functions())
    assert used <= declared, sorted(used - declared)
    # what a drop-in needs: both LcEncoding implementors, generic over the field, and the commit object
    assert re.search(r"impl<Ft, Rn, Rd> LcEncoding for HipLigeroEncodingRho<Ft, Rn, Rd>", hi)
    assert re.search(r"impl<Ft, S> LcEncoding for HipSdigEncodingS<Ft, S>", hi)
    for f in ("ft63::Ft63", "ft127::Ft127", "ft191::Ft191", "ft255::Ft255"):
        assert "unsafe impl HipField for lcpc_test_fields::" + f in hi
    for sym in ("lcpc_ctx_create", "lcpc_encode_rows", "lcpc_dims_ok", "lcpc_get_n_col_opens", "lcpc_get_n_degree_tests", "lcpc_commit",
                "lcpc_commit_device", "lcpc_get_root", "lcpc_prove", "lcpc_verify", "lcpc_commit_bincode_write", "lcpc_commit_from_bincode",
                "lcpc_comm_init", "lcpc_commit_sharded_device", "lcpc_prove_sharded_rccl", "lcpc_static_get_dims_ml"):
        assert sym in used, sym
    # the labels are the reference's (lcpc-2d/src/macros.rs:31-34 through def_labels!): same literal identifiers as its implementors
    assert "def_labels!(ligero_pc)" in hi and "def_labels!(sdig_pc)" in hi
    # constant names used through sys:: exist there
    sysc = set(re.findall(r"pub const (LCPC_\w+)", open(SYS).read()))
    assert set(re.findall(r"sys::(LCPC_\w+)", hi)) <= sysc


def test_rust_sources_are_lexically_balanced():
    """no rustc here: at least every bracket of the two crates closes in the right order, outside comments, strings and chars
    (a dropped brace is the commonest way a hand-written file stops compiling), and Cargo manifests name the paths that exist"""
    base = os.path.join(ROOT, "bindings", "rust")
    for rel in ("lcpc-hip-sys/src/lib.rs", "lcpc-hip-sys/build.rs", "lcpc-hip/src/lib.rs", "lcpc-hip/examples/commit_prove.rs"):
        src = open(os.path.join(base, rel), encoding="utf8").read()
        stack, i, n = [], 0, len(src)
        pairs = {")": "(", "]": "[", "}": "{"}
        while i < n:
            ch = src[i]
            if src.startswith("//", i):
                i = src.find("\n", i)
                i = n if i < 0 else i
                continue
            if src.startswith("/*", i):
                i = src.find("*/", i) + 2
                continue
            if ch == '"':
                i += 1
                while src[i] != '"':
                    i += 2 if src[i] == "\\" else 1
                i += 1
                continue
            if ch == "b" and src.startswith('b"', i):
                i += 1
                continue
            if ch == "'":
                # a char literal ('x', '\n') or a lifetime ('a, 'static, '_): literals close within 4 characters
                m = re.match(r"'(\\.|[^\\'])'", src[i:])
                i += len(m.group(0)) if m else 1
                continue
            if ch in "([{":
                stack.append((ch, src.count("\n", 0, i) + 1))
            elif ch in ")]}":
                assert stack and stack[-1][0] == pairs[ch], "%s: unbalanced %r at line %d" % (rel, ch, src.count("\n", 0, i) + 1)
                stack.pop()
            i += 1
        assert not stack, "%s: unclosed %r from line %d" % (rel, stack[-1][0], stack[-1][1])
    for crate in ("lcpc-hip-sys", "lcpc-hip"):
        man = open(os.path.join(base, crate, "Cargo.toml")).read()
        assert re.search(r'^name = "%s"' % crate, man, flags=re.M)
    ws = open(os.path.join(base, "Cargo.toml")).read()
    assert '"lcpc-hip-sys"' in ws and '"lcpc-hip"' in ws
    assert 'path = "../lcpc-hip-sys"' in open(os.path.join(base, "lcpc-hip", "Cargo.toml")).read()


# ---- the wrapper crate against the reference's own signatures -------------------------------------------------------------------
# Which reference item each public function of lcpc-hip stands in for, and how the reference's type names read on this side
# (the only differences a caller sees: the digest is fixed to BLAKE3, the transcript is the library's -- merlin's STROBE state is
# private and cannot be handed across -- and the encoder's error type is HipError).
MIRRORS = {
    # reference item                       (impl block in lcpc-hip/src/lib.rs, fn name)
    "LcCommit::commit": ("HipCommit", "commit"),
    "LcCommit::prove": ("HipCommit", "prove"),
    "LcCommit::get_root": ("HipCommit", "get_root"),
    "LcCommit::get_n_rows": ("HipCommit", "get_n_rows"),
    "LcCommit::get_n_per_row": ("HipCommit", "get_n_per_row"),
    "LcCommit::get_n_cols": ("HipCommit", "get_n_cols"),
    "LcEvalProof::verify": (None, "verify_on_device"),           # free function: `self` becomes the proof's bincode bytes
    "LigeroEncodingRho::new": ("HipLigeroEncodingRho", "new"),
    "LigeroEncodingRho::new_ml": ("HipLigeroEncodingRho", "new_ml"),
    "LigeroEncodingRho::new_from_dims": ("HipLigeroEncodingRho", "new_from_dims"),
    "SdigEncodingS::new": ("HipSdigEncodingS", "new"),
    "SdigEncodingS::new_ml": ("HipSdigEncodingS", "new_ml"),
    "SdigEncodingS::new_from_dims": ("HipSdigEncodingS", "new_from_dims"),
}
TYPE_MAP = [(r"\bFldT<E>", "E::F"), (r"\bErrT<E>", "HipError"), (r"\bTranscript\b", "HipTranscript"), (r"\bD\b", "Blake3"),
            (r"\bProverResult<(.*), HipError>$", r"Result<\1, ProverError<HipError>>"),
            (r"\bVerifierResult<(.*), HipError>$", r"Result<\1, VerifierError<HipError>>")]


def norm_type(t, ours=False):
    t = " ".join(t.split())
    if ours:
        t = re.sub(r"&'a ", "&", t)                  # HipCommit<'a, E> borrows its encoder for its lifetime
    else:
        for pat, rep in TYPE_MAP:
            t = re.sub(pat, rep, t)
    return t.replace(" ", "")


def crate_functions():
    """(impl type or None, fn name) -> {"args": [[name, type]], "ret": type} for every `pub fn` of lcpc-hip/src/lib.rs"""
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_reference_api", os.path.join(ROOT, "tests", "golden", "make_reference_api.py"))
    M = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(M)
    src = strip_rs(open(HI).read())
    src = src.split("#[cfg(test)]")[0]
    out = {}
    spans = []
    for m in re.finditer(r"impl(?:<[^{]*?>)?\s+(\w+)(?:<[^{]*?>)?\s*(?:where[^{]*)?\{", src):
        if " for " in m.group(0):
            continue
        (start, body), = M.block_of(src[m.start():], r"impl(?:<[^{]*?>)?\s+%s\b[^{;]*?\{" % m.group(1))[:1]
        spans.append((m.start(), m.start() + len(body)))
        for f in re.finditer(r"pub\s+(?:unsafe\s+)?fn\s+(\w+)\s*(<[^(]*?>)?\s*\((.*?)\)\s*(?:->\s*([^{;]+?))?\s*(?:where[^{;]*)?\{", body, flags=re.S):
            a = []
            for x in M.split_args(f.group(3)):
                a.append(["self", x] if x in ("&self", "self", "&mut self") else [x.split(":", 1)[0].strip(), " ".join(x.split(":", 1)[1].split())])
            out.setdefault((m.group(1), f.group(1)), {"args": a, "ret": " ".join(f.group(4).split()) if f.group(4) else None})
    for f in re.finditer(r"\npub\s+fn\s+(\w+)\s*(<[^(]*?>)?\s*\((.*?)\)\s*(?:->\s*([^{;]+?))?\s*(?:where[^{;]*)?\{", src, flags=re.S):
        a = [[x.split(":", 1)[0].strip(), " ".join(x.split(":", 1)[1].split())] for x in M.split_args(f.group(3))]
        out[(None, f.group(1))] = {"args": a, "ret": " ".join(f.group(4).split()) if f.group(4) else None}
    return out, M


def test_wrapper_signatures_equal_reference():
    """every public function of lcpc-hip that stands in for a reference item takes the reference's arguments -- same names, same
    order, same types after the three documented substitutions -- and returns the reference's type (lcpc-2d/src/lib.rs:270-312,
    518-527; lcpc-ligero-pc/src/lib.rs:121-148; lcpc-brakedown-pc/src/lib.rs:103-137).  The reference side is the committed
    table tests/golden/reference_api.json (made by tests/golden/make_reference_api.py); where /root/reference is present the
    table itself is re-derived and must be current."""
    import json
    ref = json.load(open(os.path.join(ROOT, "tests", "golden", "reference_api.json")))
    ours, M = crate_functions()
    if os.path.isdir(M.REF):
        assert M.build() == ref, "tests/golden/reference_api.json is stale: python tests/golden/make_reference_api.py"
    for item, key in MIRRORS.items():
        assert key in ours, "lcpc-hip has no %s::%s" % key
        r, o = ref[item], ours[key]
        ra = [a for a in r["args"]]
        oa = [a for a in o["args"]]
        if item == "LcEvalProof::verify":
            assert oa[0] == ["proof_bytes", "&[u8]"]               # `&self` of the proof
            ra, oa = ra[1:], oa[1:]
        assert [a[0] for a in ra] == [a[0] for a in oa], "%s: argument names %s vs %s" % (item, [a[0] for a in ra], [a[0] for a in oa])
        for (n, rt), (_, ot) in zip(ra, oa):
            assert norm_type(rt) == norm_type(ot, True), "%s, argument %s: reference %s, crate %s" % (item, n, rt, ot)
        assert norm_type(r["ret"]) == norm_type(o["ret"], True), "%s returns %s, the crate's %s" % (item, r["ret"], o["ret"])
    # the trait the encoders implement: method names and arities of LcEncoding (the trait's own signatures are the compiler's to check)
    src = strip_rs(open(HI).read())
    for enc in ("HipLigeroEncodingRho", "HipSdigEncodingS"):
        blk = re.search(r"impl<[^{]*?>\s+LcEncoding\s+for\s+%s\b.*?\n\}" % enc, src, flags=re.S).group(0)
        for name in ("encode", "get_dims", "dims_ok", "get_n_col_opens", "get_n_degree_tests"):
            m = re.search(r"fn\s+%s\s*(?:<[^(]*?>)?\s*\((.*?)\)" % name, blk, flags=re.S)
            assert m, "%s: LcEncoding::%s missing" % (enc, name)
            assert len(M.split_args(m.group(1))) == len(ref["LcEncoding::" + name]["args"]), (enc, name)


def test_ci_script_and_readme_state_the_pin():
    """bindings/rust/ci.sh holds the three commands whose first green run pins the oracle to the real crates; the README says so
    and says that the crates have never been compiled here"""
    sh = open(os.path.join(ROOT, "bindings", "rust", "ci.sh")).read()
    for needle in ("cargo test -p lcpc-hip", "--example commit_prove", "oracle/repin", "compare.py"):
        assert needle in sh, needle
    rd = open(os.path.join(ROOT, "bindings", "rust", "README.md")).read().lower()
    assert "never been compiled" in rd and "ci.sh" in rd and "partial" in rd
