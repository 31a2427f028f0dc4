#!/usr/bin/env python3
"""tests/golden/make_reference_api.py -> tests/golden/reference_api.json

The public signatures of the reference items the Rust wrapper crate (bindings/rust/lcpc-hip) mirrors, parsed from
/root/reference: for each item the argument NAMES and TYPES in order and the return type -- data about the API surface, no source
text.  tests/test_rust_bindings.py compares the crate's functions with this table (and, where /root/reference is present, checks that
the table is current).

  LcCommit::{commit, prove, get_root, get_n_rows, get_n_per_row, get_n_cols}      lcpc-2d/src/lib.rs:270-312
  LcEvalProof::verify                                                             lcpc-2d/src/lib.rs:518-527
  LigeroEncodingRho::{new, new_ml, new_from_dims}                                 lcpc-ligero-pc/src/lib.rs:121-148
  SdigEncodingS::{new, new_ml, new_from_dims}                                     lcpc-brakedown-pc/src/lib.rs:103-137
  LcEncoding::{encode, get_dims, dims_ok, get_n_col_opens, get_n_degree_tests}    lcpc-2d/src/lib.rs:74-104
"""
import json
import os
import re
import sys

REF = "/root/reference"
ITEMS = [
    ("lcpc-2d/src/lib.rs", "LcCommit", ["commit", "prove", "get_root", "get_n_rows", "get_n_per_row", "get_n_cols"]),
    ("lcpc-2d/src/lib.rs", "LcEvalProof", ["verify"]),
    ("lcpc-ligero-pc/src/lib.rs", "LigeroEncodingRho", ["new", "new_ml", "new_from_dims"]),
    ("lcpc-brakedown-pc/src/lib.rs", "SdigEncodingS", ["new", "new_ml", "new_from_dims"]),
    ("lcpc-2d/src/lib.rs", "LcEncoding", ["encode", "get_dims", "dims_ok", "get_n_col_opens", "get_n_degree_tests"]),
]


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
    return [" ".join(a.split()) for a in out if a.strip()]


def block_of(src, header_re):
    """the brace-balanced bodies of every `impl ... Name` / `trait Name` block"""
    out = []
    for m in re.finditer(header_re, src):
        i = src.index("{", m.end() - 1) if src[m.end() - 1] != "{" else m.end() - 1
        depth, j = 0, i
        while True:
            if src[j] == "{":
                depth += 1
            elif src[j] == "}":
                depth -= 1
                if depth == 0:
                    break
            j += 1
        out.append((m.start(), src[i + 1:j]))
    return out


def signatures(path, type_name, names, root=REF):
    src = open(os.path.join(root, path)).read()
    src = re.sub(r"//[^\n]*", "", src)
    blocks = block_of(src, r"(?:impl(?:<[^{]*?>)?\s+%s\b[^{;]*?\{|pub\s+trait\s+%s\b[^{]*?\{)" % (type_name, type_name))
    found = {}
    for _, body in blocks:
        for m in re.finditer(r"(?:pub\s+)?fn\s+(\w+)\s*(<[^(]*?>)?\s*\((.*?)\)\s*(?:->\s*([^{;]+?))?\s*(?:where[^{;]*)?[{;]", body, flags=re.S):
            name, args, ret = m.group(1), m.group(3), m.group(4)
            if name not in names or name in found:
                continue
            a = []
            for x in split_args(args):
                if x in ("&self", "self", "&mut self"):
                    a.append(["self", x])
                else:
                    n, t = x.split(":", 1)
                    a.append([n.strip(), " ".join(t.split())])
            found[name] = {"args": a, "ret": " ".join(ret.split()) if ret else None}
    missing = [n for n in names if n not in found]
    assert not missing, (path, type_name, missing)
    return found


def build(root=REF):
    return {"%s::%s" % (t, n): sig for path, t, names in ITEMS for n, sig in signatures(path, t, names, root).items()}


if __name__ == "__main__":
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_api.json")
    api = build()
    with open(out, "w") as f:        # one item per line
        f.write("{\n" + ",\n".join(" %s: %s" % (json.dumps(k), json.dumps(api[k], sort_keys=True)) for k in sorted(api)) + "\n}\n")
    print("wrote", out, file=sys.stderr)
