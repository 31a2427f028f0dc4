#!/usr/bin/env python3
"""lcpc_amd/csrc/gen/gen_wmul_asm.py > lcpc_amd/csrc/field_wmul_gen.h

The "shifted multiples" multiply of the row NTT kernels (ntt_l9s.hip, ntt_lns.hip): x * w mod p for a WAVE-UNIFORM w whose N constants
W_j = balanced(w * 2^(W j) mod p), j = 0..N-1, sit in memory as N^2 dwords t = N k + j (limb k of W_j; limbs 0..N-2 in [0, 2^W), the top
limb signed), for the four test fields of /root/reference/lcpc-test-fields/src/lib.rs:13-59 in the limb forms of field_dev.h (Ft255:
9 x 29 bits) and field_ln.h (Ft63 3 x 26, Ft127 5 x 29, Ft191 7 x 29):

    z = sum_j x_j * W_j                       N^2 v_mad_i64_i32, the W limbs as SGPR operands (s_load inside the statement)
    q' ~ z / 2p from the two top columns      5 instructions
    r = z - (q' + 1) * 2p                     N v_mad_i64_i32 folded into the column chain
    -> r_0..r_(N-2) in [0, 2^W), r_(N-1) signed, r == x * w (mod p), r in [-2.5p, 1.6p) for every field (tests/wmul_sim.py wmul_bounds)
Preconditions: sum_j |x_j| < N * 2^W (limbs of a normalised value, or of a difference of two); W limbs as above.

against the N^2 + N (N - 1) mads + 4 N of the Montgomery form with per-lane twiddles (gen_r29_asm.py, gen_ln_asm.py).  The multiplies
twiddle-by-twiddle are the reference's fft_io_pc butterflies (lcpc-ligero-pc/src/lib.rs:162-164); exact arithmetic mod p.

As a module: FIELDS, params(field), build(field) -> the instruction list; tests/wmul_sim.py runs that very list on Python integers with
the instructions' own wrap-around semantics and states the output bounds (tests/test_gen_wmul.py)."""

# name, modulus, N limbs, W bits, VGPR budget of the kernel that inlines it (the fixed registers sit at its top)
FIELDS = {
    "ft255": (0x663c799b6e4d2900fda9df04b9575969ef73c79086595f3002a4f20000000001, 9, 29, 128),
    "ft63": (0x46d0760000000001, 3, 26, 64),
    "ft127": (0x6e754097ba20e0bf7f2bd90000000001, 5, 29, 72),
    "ft191": (0x453708aa3fbc8dda936888270ceecbcdd246820000000001, 7, 29, 96),
}


def params(field):
    P, N, W, VB = FIELDS[field]
    B = 1 << (W * (N - 1))
    ptop = P // B
    zmax = N * (1 << W) * (ptop + 1) // 2 + N + 1        # |Z| = |col_(N-1) + (col_(N-2) >> W)|
    s1 = max(0, zmax.bit_length() - 31)                  # (Z >> s1) fits 31 bits + sign
    if field == "ft255":
        s2 = 31                                          # (round 5's constants, kept: the K1s kernels were measured with them)
    else:
        s2 = 31
        while ((1 << (s1 + s2)) * B) // (2 * P) < (1 << 31):
            s2 += 1
        s2 -= 1                                          # the largest shift whose multiplier still fits 31 bits
    MU = ((1 << (s1 + s2 - 1)) * B) // P                 # q' = ((Z >> s1) * MU) >> s2 ~ z / 2p
    assert MU < (1 << 31) and s2 < 64
    return P, N, W, VB, s1, s2, MU


def sreg_map(field):
    """SGPR of W limb t = N k + j.  s[16:31], s[36:99] and s34 (s32 / s33 / s100+ are reserved by the ABI)."""
    P, N, W, VB, s1, s2, MU = params(field)
    if field == "ft255":
        return (lambda t: 16 + t if t < 16 else (36 + t - 16 if t < 80 else 34)), 35, None
    # smaller fields: the table from s36 up, then MU, then the N limbs of -2p (scalar operands too: no VGPRs for constants)
    return (lambda t: 36 + t), 36 + N * N, 36 + N * N + 1


def fixed_regs(field):
    """accumulators A (main chain), B (column N-2), C (column N-1), the temp pair T of the quotient estimate, the quotient Q"""
    VB = FIELDS[field][3]
    if field == "ft255":
        return VB - 2, VB - 4, VB - 6, VB - 10, VB - 8
    return VB - 2, VB - 4, VB - 6, VB - 2, VB - 7       # T shares A's pair (the chain starts after the quotient): 7 fixed registers


def build(field):
    P, N, W, VB, s1, s2, MU = params(field)
    M = (1 << W) - 1
    sreg, S_MU, S_NP = sreg_map(field)
    A, Bv, C, T, Q = fixed_regs(field)
    acc = lambda r: "v[%d:%d]" % (r, r + 1)
    X = lambda j: "%%[x%d]" % j
    NP = (lambda k: "%%[n%d]" % k) if S_NP is None else (lambda k: "s%d" % (S_NP + k))
    R = lambda k: "%%[r%d]" % k
    Wl = lambda k, j: "s%d" % sreg(N * k + j)
    ins = []
    if field == "ft255":
        for blk in range(5):
            ins.append("s_load_dwordx16 s[%d:%d], %%[w], 0x%x" % (sreg(16 * blk), sreg(16 * blk) + 15, 64 * blk))
        ins.append("s_load_dword s%d, %%[w], 0x140" % sreg(80))
    else:
        t = 0
        while t < N * N:                                 # aligned power-of-two runs (s36 is a multiple of 4)
            n = 16
            while n > N * N - t or (sreg(t) % min(n, 4)) != 0:
                n //= 2
            ins.append(("s_load_dword s%d" % sreg(t) if n == 1 else "s_load_dwordx%d s[%d:%d]" % (n, sreg(t), sreg(t) + n - 1)) + ", %%[w], 0x%x" % (4 * t))
            t += n
        for k in range(N):
            ins.append("s_mov_b32 s%d, 0x%x" % (S_NP + k, (-(((2 * P) >> (W * k)) & (M if k < N - 1 else (1 << 32) - 1))) & 0xffffffff))
    ins.append("s_mov_b32 s%d, 0x%x" % (S_MU, MU))
    ins.append("s_waitcnt lgkmcnt(0)")
    for k, reg in ((N - 1, C), (N - 2, Bv)):             # the two top column sums, kept
        for j in range(N):
            ins.append("v_mad_i64_i32 %s, vcc, %s, %s, %s" % (acc(reg), X(j), Wl(k, j), "0" if j == 0 else acc(reg)))
    ins.append("v_ashrrev_i64 %s, %d, %s" % (acc(T), W, acc(Bv)))
    ins.append("v_add_co_u32 v%d, vcc, v%d, v%d" % (T, T, C))
    ins.append("v_addc_co_u32 v%d, vcc, v%d, v%d, vcc" % (T + 1, T + 1, C + 1))
    # Ft255: |Z| < 9 * 2^29 * 2^22.7 / 2 = 2^53.9: Z >> 23 fits 31 bits + sign (>> 22 does not: one multiply in 10^7 overflowed)
    if s1:
        ins.append("v_ashrrev_i64 %s, %d, %s" % (acc(T), s1, acc(T)))
    ins.append("v_mad_i64_i32 %s, vcc, v%d, s%d, 0" % (acc(T), T, S_MU))
    ins.append("v_ashrrev_i64 %s, %d, %s" % (acc(T), s2, acc(T)))
    ins.append("v_add_u32 v%d, 1, v%d" % (Q, T))         # the floors above lose < 2.7 units: centre the remainder
    for k in range(N):
        if k < N - 2:
            for j in range(N):
                ins.append("v_mad_i64_i32 %s, vcc, %s, %s, %s" % (acc(A), X(j), Wl(k, j), "0" if (k == 0 and j == 0) else acc(A)))
        else:
            src = Bv if k == N - 2 else C
            ins.append("v_add_co_u32 v%d, vcc, v%d, v%d" % (A, A, src))
            ins.append("v_addc_co_u32 v%d, vcc, v%d, v%d, vcc" % (A + 1, A + 1, src + 1))
        ins.append("v_mad_i64_i32 %s, vcc, v%d, %s, %s" % (acc(A), Q, NP(k), acc(A)))
        if k < N - 1:
            ins.append("v_and_b32 %s, 0x%x, v%d" % (R(k), M, A))
            ins.append("v_ashrrev_i64 %s, %d, %s" % (acc(A), W, acc(A)))
        else:
            ins.append("v_mov_b32 %s, v%d" % (R(k), A))
    return ins


def emit(field):
    P, N, W, VB, s1, s2, MU = params(field)
    sreg, S_MU, S_NP = sreg_map(field)
    ins = build(field)
    A = fixed_regs(field)[0]
    T = min(fixed_regs(field))
    n_valu = sum(1 for i in ins if i.startswith("v_"))
    out = []
    if field == "ft255":
        out.append("// GENERATED by lcpc_amd/csrc/gen/gen_wmul_asm.py -- do not edit.  %d VALU instructions (%d mads) per multiply." % (n_valu, sum(1 for i in ins if i.startswith("v_mad"))))
        out.append("// np2: the nine limbs of -2p (signed); w: wave-uniform pointer to the 81 dwords of one twiddle's shifted multiples.")
        out.append("#define WMUL_MU 0x%xu" % MU)
        out.append("LCPC_DEV void wmul_u(const u32* x, const u32* np2, const u32* w, u32* r) {")
    else:
        out.append("")
        out.append("// %s: %d limbs of %d bits, %d VALU instructions (%d mads); w: wave-uniform pointer to the %d dwords t = %d k + j of one twiddle's" % (field, N, W, n_valu, sum(1 for i in ins if i.startswith("v_mad")), N * N, N))
        out.append("// shifted multiples.  Fixed registers v[%d:%d] (the top of the %d-VGPR budget of ntt_pass_lns_kernel<%s>)." % (T, A + 1, VB, field))
        out.append("LCPC_DEV void wmul_u_%s(const u32* x, const u32* w, u32* r) {" % field)
    out.append("  asm volatile(")
    for t in ins:
        out.append('      "%s\\n\\t"' % t)
    outs = ", ".join('[r%d] "=&v"(r[%d])' % (k, k) for k in range(N))
    inps = ['[x%d] "v"(x[%d])' % (j, j) for j in range(N)]
    if S_NP is None:
        inps += ['[n%d] "v"(np2[%d])' % (k, k) for k in range(N)]
    inps += ['[w] "s"(w)']
    sregs = ['"s%d"' % sreg(t) for t in range(N * N)] + ['"s%d"' % S_MU]
    if S_NP is not None:
        sregs = ['"s%d"' % sreg(t) for t in range(N * N)] + ['"s%d"' % S_MU] + ['"s%d"' % (S_NP + k) for k in range(N)]
    clob = ", ".join(['"vcc"'] + ['"v%d"' % v for v in range(T, A + 2)] + sregs + ['"memory"'])
    out.append("      : %s\n      : %s\n      : %s);" % (outs, ", ".join(inps), clob))
    out.append("}")
    return out


def main():
    out = emit("ft255")
    out.append("")
    out.append("// ---- the same multiply for the limb forms of field_ln.h (K1n, ntt_lns.hip); -2p and MU as scalar constants -------------------------")
    for f in ("ft63", "ft127", "ft191"):
        out += emit(f)
    print("\n".join(out))


if __name__ == "__main__":
    main()
