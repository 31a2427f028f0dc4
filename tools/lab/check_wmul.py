#!/usr/bin/env python3
"""tools/check_wmul.py wmul_dump.txt -- host check of tools/ubench_wmul.hip's dump: 64 lanes, x (9 limbs) -> after three uniform-twiddle
multiplies; r must be == x * w0 * w1 * w2 (mod p) with |r| < 5p and limbs 0..7 in [0, 2^29)."""
import sys
P = 0x663c799b6e4d2900fda9df04b9575969ef73c79086595f3002a4f20000000001
ws, ok, worst = [], 0, 0.0
for line in open(sys.argv[1]):
    t = line.split()
    if t[0] == "w":
        ws.append(int(t[1], 16))
    else:
        v = [int(a) for a in t[1:]]
        x = sum(l << (29 * i) for i, l in enumerate(v[:9]))
        r = sum(l << (29 * i) for i, l in enumerate(v[9:]))
        want = x
        for w in ws:
            want = want * w % P
        assert (r - want) % P == 0, ("wrong residue", v)
        assert all(0 <= l < (1 << 29) for l in v[9:17]), ("limb range", v)
        worst = max(worst, abs(r) / P)
        ok += 1
print("wmul check: %d lanes ok, max |r| / p = %.3f" % (ok, worst))
