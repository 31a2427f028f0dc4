"""tests/wmul_sim.py -- the generated shifted-multiples multipliers of lcpc_amd/csrc/gen/gen_wmul_asm.py (field_wmul_gen.h: the multiply by a
wave-uniform twiddle in the row NTT kernels, the butterflies of /root/reference/lcpc-ligero-pc/src/lib.rs:162-164) as Python integers:
simulate() interprets the generator's own instruction list with 32 / 64-bit wrap-around, shifted_multiples() builds the table the
kernels read, wmul_bounds() is the proved range of the result.  Test infrastructure (tests/test_gen_wmul.py)."""
import os
import re
import sys
from fractions import Fraction as Fr

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "lcpc_amd", "csrc", "gen"))
from gen_wmul_asm import FIELDS, build, params, sreg_map  # noqa: E402,F401


def wmul_bounds(field):
    """(lo, hi) in units of p with lo <= r / p < hi for every admissible input.  With qt = z / 2p the true quotient:
    Z = col_(N-1) + floor(col_(N-2) / 2^W) misses z / B by less than N + 2 units, i.e. (N + 2) / (2 p / B) of a quotient unit ("small");
    the floor of Z >> s1 loses [0, 2^s1 / (2 p / B)); MU = floor(mu) loses |Z >> s1| (mu - MU) / 2^s2 =: e towards zero; the last
    floor loses [0, 1) and the + 1 centres it:  qt - Q in [-1 - small - e, small + 2^s1 / (2 p / B) + e),  r = (qt - Q) 2p."""
    P, N, W, VB, s1, s2, MU = params(field)
    B = 1 << (W * (N - 1))
    ptop = Fr(P, B)
    frac = Fr((1 << (s1 + s2 - 1)) * B, P) - MU
    zs = (Fr(N * (1 << W), 2) * ptop + N + 1) / (1 << s1) + 1
    e = zs * frac / (1 << s2)
    small = Fr(N + 2) / (2 * ptop)
    return -2 * (1 + small + e), 2 * (small + Fr(1 << s1) / (2 * ptop) + e)


def shifted_multiples(field, w):
    """w: the plain residue.  -> N^2 words t = N k + j (u32, two's complement)."""
    P, N, W, VB, s1, s2, MU = params(field)
    M = (1 << W) - 1
    tab = [0] * (N * N)
    for j in range(N):
        v = (w << (W * j)) % P
        if v > (P - 1) // 2:
            v -= P
        for k in range(N):
            tab[N * k + j] = ((v >> (W * k)) & M) if k < N - 1 else ((v >> (W * k)) & 0xffffffff)
    return tab


def simulate(field, x, wtab, ins=None):
    """x: N signed limb values (python ints, |x_j| < 2^31); wtab: shifted_multiples().  Returns the N result limbs as signed ints
    (limbs 0..N-2 in [0, 2^W), the top limb two's complement), having run build(field) with 32 / 64-bit wrap-around."""
    P, N, W, VB, s1, s2, MU = params(field)
    sreg, S_MU, S_NP = sreg_map(field)
    ins = ins or build(field)
    s32 = lambda v: ((v + (1 << 31)) & 0xffffffff) - (1 << 31)
    s64 = lambda v: ((v + (1 << 63)) & ((1 << 64) - 1)) - (1 << 63)
    vg, sg, named = {}, {}, {}
    for j in range(N):
        named["x%d" % j] = x[j] & 0xffffffff
    if S_NP is None:
        for k in range(N):
            lim = ((2 * P) >> (W * k)) & (((1 << W) - 1) if k < N - 1 else 0xffffffff)
            named["n%d" % k] = (-lim) & 0xffffffff
    vcc = 0

    def rd(op):
        op = op.strip()
        if op.startswith("%["):
            return named[op[2:-1]]
        if op.startswith("v["):
            lo = int(op[2:op.index(":")])
            return vg.get(lo, 0) | (vg.get(lo + 1, 0) << 32)
        if op.startswith("v"):
            return vg.get(int(op[1:]), 0)
        if op.startswith("s"):
            return sg[int(op[1:])]
        return int(op, 0) & 0xffffffff

    def wr(op, val, wide=False):
        op = op.strip()
        if op.startswith("%["):
            named[op[2:-1]] = val & 0xffffffff
        elif wide:
            lo = int(op[2:op.index(":")])
            vg[lo], vg[lo + 1] = val & 0xffffffff, (val >> 32) & 0xffffffff
        else:
            vg[int(op[1:])] = val & 0xffffffff

    for line in ins:
        mn, rest = line.split(" ", 1)
        ops = [o.strip() for o in re.split(r",\s*(?![^\[]*\])", rest)]
        if mn.startswith("s_load_dword"):
            n = 1 if mn == "s_load_dword" else int(mn[len("s_load_dwordx"):])
            base = int(ops[0][1:]) if n == 1 else int(ops[0][2:ops[0].index(":")])
            off = int(ops[2], 0) // 4
            for i in range(n):
                sg[base + i] = wtab[off + i] if off + i < len(wtab) else 0
        elif mn == "s_mov_b32":
            sg[int(ops[0][1:])] = int(ops[1], 0) & 0xffffffff
        elif mn == "s_waitcnt":
            pass
        elif mn == "v_mad_i64_i32":
            a, b = s32(rd(ops[2])), s32(rd(ops[3]))
            c = 0 if ops[4] == "0" else s64(rd(ops[4]))
            wr(ops[0], (a * b + c) & ((1 << 64) - 1), True)
        elif mn == "v_ashrrev_i64":
            wr(ops[0], (s64(rd(ops[2])) >> int(ops[1])) & ((1 << 64) - 1), True)
        elif mn == "v_add_co_u32":
            t = rd(ops[2]) + rd(ops[3])
            vcc = t >> 32
            wr(ops[0], t)
        elif mn == "v_addc_co_u32":
            t = rd(ops[2]) + rd(ops[3]) + vcc
            vcc = t >> 32
            wr(ops[0], t)
        elif mn == "v_and_b32":
            wr(ops[0], rd(ops[1]) & rd(ops[2]))
        elif mn == "v_add_u32":
            wr(ops[0], rd(ops[1]) + rd(ops[2]))
        elif mn == "v_mov_b32":
            wr(ops[0], rd(ops[1]))
        else:
            raise ValueError(line)
    return [named["r%d" % k] if k < N - 1 else s32(named["r%d" % k]) for k in range(N)]
