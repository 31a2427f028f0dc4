"""The generated shifted-multiples multipliers (lcpc_amd/csrc/gen/gen_wmul_asm.py -> field_wmul_gen.h: the multiply by a wave-uniform
twiddle of the row NTT, /root/reference/lcpc-ligero-pc/src/lib.rs:162-164 butterflies) run as Python integers: the generator's own
instruction list, interpreted with the instructions' 32 / 64-bit wrap-around, must give x * w (mod p), normalised limbs and a value
inside wmul_bounds() for every admissible input shape of the four test fields (lcpc-test-fields/src/lib.rs:13-59).  CPU only."""
import os
import random
import sys

import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import wmul_sim as G  # noqa: E402


def _limbs(v, N, W):
    return [(v >> (W * k)) & ((1 << W) - 1) for k in range(N - 1)] + [v >> (W * (N - 1))]


@pytest.mark.parametrize("field", list(G.FIELDS))
def test_wmul_simulated(field):
    P, N, W, VB, s1, s2, MU = G.params(field)
    ins = G.build(field)
    lo, hi = G.wmul_bounds(field)
    assert -2.5 <= lo and hi <= 1.6
    rng = random.Random(0xC0FFEE + N)
    ptop = P >> (W * (N - 1))
    seen_lo, seen_hi = 0.0, 0.0
    for it in range(1500):
        w = rng.randrange(1, P) if it % 7 else rng.choice([1, P - 1, (P - 1) // 2, (P + 1) // 2, 2, P - 2])
        tab = G.shifted_multiples(field, w)
        mode = it % 5
        if mode == 0:      # a normalised value, |value| < 4p
            x = _limbs(rng.randrange(-4 * P, 4 * P), N, W)
        elif mode == 1:    # the difference of two
            x = [a - b for a, b in zip(_limbs(rng.randrange(-4 * P, 4 * P), N, W), _limbs(rng.randrange(-4 * P, 4 * P), N, W))]
        elif mode == 2:    # every limb at its extreme, one sign
            sg = rng.choice([1, -1])
            x = [sg * ((1 << W) - 1)] * (N - 1) + [sg * rng.randrange(0, 8 * ptop)]
        elif mode == 3:    # random signs
            x = [rng.choice([-1, 1]) * rng.randrange(0, 1 << W) for _ in range(N - 1)] + [rng.randrange(-8 * ptop, 8 * ptop)]
        else:              # extremes of either sign and zeros
            x = [rng.choice([-(1 << W) + 1, (1 << W) - 1, 0]) for _ in range(N - 1)] + [rng.choice([-1, 1]) * 8 * ptop]
        assert sum(abs(l) for l in x) < N * (1 << W)
        xv = sum(l << (W * k) for k, l in enumerate(x))
        r = G.simulate(field, x, tab, ins)
        rv = sum(l << (W * k) for k, l in enumerate(r))
        assert (rv - xv * w) % P == 0
        assert all(0 <= l < (1 << W) for l in r[:-1])
        assert lo * P <= rv < hi * P
        seen_lo, seen_hi = min(seen_lo, rv / P), max(seen_hi, rv / P)
    assert seen_lo < -1.5          # (the sweep does reach the neighbourhood of the lower bound)


def test_wmul_header_is_the_generators():
    """the four multipliers in the built header are the ones simulated above"""
    hdr = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "lcpc_amd", "csrc", "field_wmul_gen.h")
    if not os.path.exists(hdr):
        pytest.skip("header not generated yet (lcpc_amd/csrc/Makefile makes it)")
    text = open(hdr).read()
    for f in G.FIELDS:
        for line in G.build(f):
            assert '"%s\\n\\t"' % line in text
