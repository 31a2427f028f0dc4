#!/usr/bin/env python3
"""oracle/repin/compare.py -- `cargo +nightly run --release | python compare.py`: checks every line the Rust dumper prints
against tests/golden/commit_cases.json (the fixture made by the Python restatement).  Exit 0 = the oracle is pinned."""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
cases = {c["name"]: c for c in json.load(open(os.path.join(HERE, "..", "..", "tests", "golden", "commit_cases.json")))}
bad = n = 0
for line in sys.stdin:
    t = line.split()
    if len(t) < 3 or t[0] not in cases:
        continue
    c, key = cases[t[0]], t[1]
    if key == "dims":
        want = [str(c["n_rows"]), str(c["n_per_row"]), str(c["n_cols"])]
        got = t[2:5]
    elif key == "n_col_opens":
        want, got = [str(c["n_col_opens"]), str(c["n_degree_tests"])], [t[2], t[4]]
    elif key == "commit_bincode_len":
        want, got = [str(c["commit_bincode_len"]), c["commit_bincode_sha256"]], [t[2], t[4]]
    elif key == "proof_len":
        want, got = [str(c["proof_len"]), c["proof_sha256"], c["proof_blake3"]], [t[2], t[4], t[6]]
    else:
        want, got = [str(c[key])], [t[2]]
    n += 1
    if want != got:
        bad += 1
        print("MISMATCH %s %s: reference %s, fixture %s" % (t[0], key, got, want))
print("%d lines compared, %d mismatches" % (n, bad))
sys.exit(1 if bad or n == 0 else 0)
