// lcpc_amd/csrc/field_ln.h -- lazy signed reduced-radix arithmetic for Ft63 / Ft127 / Ft191 (the row NTT of
// ntt_lns.hip): the scheme of field_dev.h's namespace l9 (Ft255: 9 limbs of 29 bits) for the other three test fields of
// /root/reference/lcpc-test-fields/src/lib.rs:18-58 (ff_derive [3P]: Montgomery form, R = 2^(64 L)).
//
//   field   N limbs x W bits   R' = 2^(N W)   spare bits N W - log2 p   top limb of p (PTOP)
//   Ft63    3 x 26             2^78           15.9                      0x46d     (p >> 52)
//   Ft127   5 x 29             2^145          18.2                      0x6e7     (p >> 116)
//   Ft191   7 x 29             2^203          12.9                      0x8a6e    (p >> 174)
// (26-bit limbs for Ft63: with 29 the top limb of p would be 17, too coarse for the quotient estimate of the clamp.)
//
// An element between stages: N limbs, limbs 0..N-2 in [0, 2^W), top limb two's complement; |value| < 4p; value == the true
// value (mod p) ("invariant I", as in l9).  add / sub are N plain limb operations; the Montgomery multiply
// (field_ln_gen.h: ONE asm statement, N^2 + N(N-1) v_mad_i64_i32 on a single 64-bit accumulator, negative quotient
// digits) accepts limbs in (-2^(W+1), 2^(W+1)), |value| < 16p, and returns a normalised value in (-p - eps, eps],
// eps = 16 p^2 / R' < p / 400; sums of sums are brought back to [0, p + 64 B), B = 2^(W(N-1)), by the quotient-estimate
// clamp.  Data stays in ff_derive's R = 2^(32 NL) form in HBM: the twiddles are pre-scaled to w^i * R' mod p, so that
// REDC_R'(a R * w R') = (a w) R.  Exact reduction to [0, p) happens once per element, at the last pass's store.
#pragma once
#include "field_dev.h"

namespace lcpc {

template <int FID> struct LnField;
template <> struct LnField<FT63> {
  static constexpr int FID = FT63, N = 3, W = 26, NL = 2, WAVES = 8, STRIDE = 4;
  static constexpr u32 limb(int k) { return (u32)(((((u64)Mod<2>::P[1] << 32) | Mod<2>::P[0]) >> (26 * k)) & ((1u << 26) - 1)); }
};
template <> struct LnField<FT127> {
  static constexpr int FID = FT127, N = 5, W = 29, NL = 4, WAVES = 7, STRIDE = 8;
  static constexpr u32 limb(int k) {
    const int b = 29 * k, w = b / 32, sh = b % 32;
    const u64 lo = Mod<4>::P[w], hi = (w + 1 < 4) ? Mod<4>::P[w + 1] : 0;
    return (u32)(((lo | (hi << 32)) >> sh) & ((1u << 29) - 1));
  }
};
template <> struct LnField<FT191> {
  static constexpr int FID = FT191, N = 7, W = 29, NL = 6, WAVES = 5, STRIDE = 8;
  static constexpr u32 limb(int k) {
    const int b = 29 * k, w = b / 32, sh = b % 32;
    const u64 lo = Mod<6>::P[w], hi = (w + 1 < 6) ? Mod<6>::P[w + 1] : 0;
    return (u32)(((lo | (hi << 32)) >> sh) & ((1u << 29) - 1));
  }
};

#include "field_ln_gen.h"   // ln_mul1s_ft63 / _ft127 / _ft191

template <int N> struct LN {
  u32 v[N];       // two's complement; the top limb (and un-normalised intermediates) may be negative
};

namespace ln {
constexpr int QOFF = 24;          // clamp table: entry i = (i - QOFF) * p, i in [0, 64)
constexpr int QBIAS = 40;         // subtracted from the top limb before the quotient estimate (keeps remainders >= 0)

// packed NL x 32 (value < 2^(32 NL)) -> N x W, normalised, non-negative.  The top limb takes every bit from W (N - 1) up:
// 12 bits for Ft63 / Ft127, 18 for Ft191.
template <class FT> LCPC_DEV LN<FT::N> from_packed(const Fe<FT::NL>& a) {
  constexpr int N = FT::N, W = FT::W, NL = FT::NL;
  LN<N> r;
#pragma unroll
  for (int k = 0; k < N; k++) {
    const int b = W * k, w = b / 32, sh = b % 32;
    u32 x;
    if (sh == 0) x = a.v[w];
    else if (w + 1 < NL) x = __builtin_amdgcn_alignbit(a.v[w + 1], a.v[w], sh);
    else x = a.v[w] >> sh;
    r.v[k] = k + 1 < N ? (x & ((1u << W) - 1)) : x;
  }
  return r;
}
// N x W (limbs 0..N-2 in [0, 2^W), top limb >= 0, value < 2^(32 NL)) -> packed
template <class FT> LCPC_DEV void to_packed(u32* out, const u32* l) {
  constexpr int N = FT::N, W = FT::W, NL = FT::NL;
#pragma unroll
  for (int w = 0; w < NL; w++) {
    u32 x = 0;
#pragma unroll
    for (int k = 0; k < N; k++) {
      const int off = W * k - 32 * w;                 // bit position of limb k inside word w
      if (off >= 0 && off < 32) x |= l[k] << off;
      else if (off < 0 && off > -32) x |= l[k] >> (-off);
    }
    out[w] = x;
  }
}
template <int N> LCPC_DEV LN<N> add(const LN<N>& a, const LN<N>& b) {
  LN<N> r;
#pragma unroll
  for (int k = 0; k < N; k++) r.v[k] = a.v[k] + b.v[k];
  return r;
}
template <int N> LCPC_DEV LN<N> sub(const LN<N>& a, const LN<N>& b) {
  LN<N> r;
#pragma unroll
  for (int k = 0; k < N; k++) r.v[k] = a.v[k] - b.v[k];
  return r;
}
// carry-propagate signed limbs (|limb| < 2^31): limbs 0..N-2 -> [0, 2^W), the top limb takes what is left (signed)
template <class FT> LCPC_DEV void normalize(LN<FT::N>& a) {
#pragma unroll
  for (int k = 0; k + 1 < FT::N; k++) {
    a.v[k + 1] += (u32)((int32_t)a.v[k] >> FT::W);
    a.v[k] &= (1u << FT::W) - 1;
  }
}
// ---- normalise + clamp in ONE carry pass (l9::clamp_q / clamp_row / clamp_apply for N limbs of W bits) -------------
// a: limbs 0..N-2 in [0, 2^31 - 4] (the sum of four normalised values), top limb signed, |value| < 16p.  With
// B = 2^(W(N-1)), V = t B + low, q = floor((t - QBIAS) / (PTOP + 1)) estimated from the UN-normalised top limb (the
// carries still in the lower limbs, c in [0, 3], are not in it yet):
//   V - q p = q (B - plow) + (rem + QBIAS + c) B + low   >= (QBIAS - |q|) B >= 0   for |q| <= 17 + 1,
//                                                         <  (PTOP + 1 + QBIAS + |q| + 1 + 3) B  <  p + 64 B.
// The row -(q p) comes from a table in LDS (the kernel negates ctx.cpp's (i - QOFF) p table when it copies it).
template <class FT> LCPC_DEV u32 clamp_q(u32 top) {
  constexpr u32 PTOP1 = FT::limb(FT::N - 1) + 1;
  constexpr int SH = 31 - __builtin_clz(PTOP1);                         // floor(log2 PTOP1)
  constexpr u32 MAGIC = (u32)((((u64)1 << (32 + SH)) + PTOP1 - 1) / PTOP1);   // ceil(2^(32+SH) / PTOP1) < 2^32
  // exact floor(n / PTOP1) while n * (MAGIC * PTOP1 - 2^(32+SH)) < 2^(32+SH): n < 64 PTOP1 here, the excess is < PTOP1
  static_assert((u64)64 * PTOP1 * PTOP1 < ((u64)1 << (32 + SH)), "magic division range");
  const u32 n = top + (u32)(QOFF * PTOP1 - QBIAS);                       // in [0, 64 PTOP1) for |value| < 16p
  return __umulhi(n, MAGIC) >> SH;
}
template <class FT> LCPC_DEV LN<FT::N> clamp_row(const u32* nqp, u32 q) {
  constexpr int N = FT::N;
  const u32* t = nqp + q * FT::STRIDE;
  LN<N> r;
  if constexpr (N == 3) {
    const uint2 a = *reinterpret_cast<const uint2*>(t);
    r.v[0] = a.x; r.v[1] = a.y; r.v[2] = t[2];
  } else {
    const uint4 a = *reinterpret_cast<const uint4*>(t);
    r.v[0] = a.x; r.v[1] = a.y; r.v[2] = a.z; r.v[3] = a.w;
    if constexpr (N == 5) r.v[4] = t[4];
    else { const uint2 b = *reinterpret_cast<const uint2*>(t + 4); r.v[4] = b.x; r.v[5] = b.y; r.v[6] = t[6]; }
  }
  return r;
}
template <class FT> LCPC_DEV void clamp_apply(LN<FT::N>& a, const LN<FT::N>& nt) {
  int32_t c = 0;
#pragma unroll
  for (int k = 0; k + 1 < FT::N; k++) {
    const int32_t d = (int32_t)(a.v[k] + nt.v[k] + (u32)c);             // in (-2^W - 1, 2^31): fits i32
    a.v[k] = (u32)d & ((1u << FT::W) - 1);
    c = d >> FT::W;
  }
  a.v[FT::N - 1] = a.v[FT::N - 1] + nt.v[FT::N - 1] + (u32)c;
}
template <class FT> LCPC_DEV LN<FT::N> mul(const LN<FT::N>& a, const LN<FT::N>& w) {
  LN<FT::N> r;
  if constexpr (FT::FID == FT63) ln_mul1s_ft63(a.v, w.v, r.v);
  else if constexpr (FT::FID == FT127) ln_mul1s_ft127(a.v, w.v, r.v);
  else ln_mul1s_ft191(a.v, w.v, r.v);
  return r;
}
// a * w mod p for a wave-uniform w given as its N shifted multiples W_j = balanced(w 2^(W j) mod p) (N^2 words t = N k + j, host:
// ctx.cpp wmul_table; field_wmul_gen.h).  a: limbs of a normalised value or of a difference of two (sum |limb| < N 2^W).  Result:
// normalised, in (-2.5p, 1.6p) (gen_wmul_asm.py wmul_bounds).  Ft127: 50 instructions against mul()'s 64, Ft191: 80 against 118.
// Ft63 has no such form: at 3 limbs the Montgomery multiply (22 instructions) is the shorter one (28).
template <class FT> constexpr bool has_mul_u = FT::N >= 5;
template <class FT> LCPC_DEV LN<FT::N> mul_u(const LN<FT::N>& a, const u32* wt) {
  static_assert(has_mul_u<FT>, "no shifted-multiples multiply for this field");
  LN<FT::N> r;
  if constexpr (FT::FID == FT127) wmul_u_ft127(a.v, wt, r.v);
  else wmul_u_ft191(a.v, wt, r.v);
  return r;
}
// ---- carry-free lazy dot product (Brakedown SpMM for Ft127 / Ft191): field_dev.h's lazy29_* for N limbs of W bits ------
// acc += x * v with x, v as N unsigned W-bit limbs (x: a packed element < p split by from_packed; v: a matrix value in the
// R'-Montgomery form, v R' mod p): N^2 v_mad_u64_u32 into 2N u64 columns, no carries.  A column receives <= N products
// < 2^(2W) per term: 6 terms fit (6 * 7 * 2^58 < 2^64) before lazy_normalize() must move the excess up; the value is
// Montgomery-reduced once per <= 60 terms: (sum + m p) / R' < 60 p^2 / R' + p < 2p since R' / p > 2^12.
template <class FT> struct LazyN {
  u64 c[2 * FT::N];
};
template <class FT> LCPC_DEV void lazy_zero(LazyN<FT>& a) {
#pragma unroll
  for (int k = 0; k < 2 * FT::N; k++) a.c[k] = 0;
}
template <class FT> LCPC_DEV void lazy_mac(LazyN<FT>& a, const LN<FT::N>& x, const LN<FT::N>& v) {
#pragma unroll
  for (int i = 0; i < FT::N; i++)
#pragma unroll
    for (int j = 0; j < FT::N; j++) a.c[i + j] += (u64)x.v[i] * v.v[j];
}
template <class FT> LCPC_DEV void lazy_normalize(LazyN<FT>& a) {
#pragma unroll
  for (int k = 0; k + 1 < 2 * FT::N; k++) {
    a.c[k + 1] += a.c[k] >> FT::W;
    a.c[k] &= (1u << FT::W) - 1;
  }
}
// value / R' mod p, fully reduced, packed
template <class FT> LCPC_DEV Fe<FT::NL> lazy_reduce(LazyN<FT>& a) {
  constexpr int N = FT::N, W = FT::W;
  constexpr u32 M = (1u << W) - 1;
  lazy_normalize<FT>(a);
  u32 m[N], r[N];
  u64 acc = 0;
#pragma unroll
  for (int k = 0; k < 2 * N; k++) {
    acc += a.c[k];
#pragma unroll
    for (int i = 0; i < N; i++) {
      const int j = k - i;
      if (i < k && j >= 1 && j < N) acc += (u64)m[i] * FT::limb(j);
    }
    if (k < N) {
      m[k] = (0u - (u32)acc) & M;          // p == 1 mod 2^W: the quotient digit is a negate-and-mask
      acc += m[k];
      acc >>= W;
    } else {
      r[k - N] = (u32)acc & M;
      acc >>= W;
    }
  }
  u32 t[FT::NL];
  to_packed<FT>(t, r);
  return fe_reduce_once<FT::NL>(t, 0u);    // < 2p < 2^(32 NL)
}
}  // namespace ln

}  // namespace lcpc
