// lcpc_amd/csrc/field_dev.h -- device-side prime-field arithmetic for gfx950 (MI355X).
//
// Replaces, on the GPU, what `#[derive(PrimeField)]` generates for the reference's four test
// fields (/root/reference/lcpc-test-fields/src/lib.rs:13-59, ff_derive [3P]): elements are
// a*R mod p with R = 2^(64 L), stored as L little-endian u64 limbs, always fully reduced (< p).
// On the device the same bytes are viewed as NL = 2L little-endian 32-bit limbs, because the
// widest integer multiplier CDNA4 has is v_mad_u64_u32 (32x32+64 -> 64).
//
// All four moduli are == 1 mod 2^32, so -p^-1 mod 2^32 = 0xffffffff and the Montgomery
// quotient digit is just a negation (m = -t0): no multiply for it, and m*p[0] is free.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace lcpc {

typedef uint32_t u32;
typedef uint64_t u64;

#define LCPC_DEV __device__ __forceinline__

// field ids match include/lcpc_hip.h
enum { FT63 = 0, FT127 = 1, FT191 = 2, FT255 = 3 };

template <int NL> struct Mod;   // modulus, 32-bit limbs, little-endian
template <> struct Mod<2> { static constexpr u32 P[2] = {0x00000001u, 0x46d07600u}; };
template <> struct Mod<4> { static constexpr u32 P[4] = {0x00000001u, 0x7f2bd900u, 0xba20e0bfu, 0x6e754097u}; };
template <> struct Mod<6> { static constexpr u32 P[6] = {0x00000001u, 0xd2468200u, 0x0ceecbcdu, 0x93688827u, 0x3fbc8ddau, 0x453708aau}; };
template <> struct Mod<8> { static constexpr u32 P[8] = {0x00000001u, 0x02a4f200u, 0x86595f30u, 0xef73c790u,
                                                        0xb9575969u, 0xfda9df04u, 0x6e4d2900u, 0x663c799bu}; };

// ---- element container: NL 32-bit limbs in registers ------------------------------------------
template <int NL> struct Fe {
  u32 v[NL];
};

template <int NL> LCPC_DEV Fe<NL> fe_zero() {
  Fe<NL> r;
#pragma unroll
  for (int i = 0; i < NL; i++) r.v[i] = 0;
  return r;
}

// global memory access: an element is NL*4 contiguous bytes (8, 16, 24 or 32), 8-byte aligned.
template <int NL> LCPC_DEV Fe<NL> fe_load(const u32* __restrict__ p) {
  Fe<NL> r;
  if constexpr (NL % 4 == 0) {
    const uint4* q = reinterpret_cast<const uint4*>(p);
#pragma unroll
    for (int i = 0; i < NL / 4; i++) {
      uint4 t = q[i];
      r.v[4 * i] = t.x; r.v[4 * i + 1] = t.y; r.v[4 * i + 2] = t.z; r.v[4 * i + 3] = t.w;
    }
  } else {
    const uint2* q = reinterpret_cast<const uint2*>(p);
#pragma unroll
    for (int i = 0; i < NL / 2; i++) {
      uint2 t = q[i];
      r.v[2 * i] = t.x; r.v[2 * i + 1] = t.y;
    }
  }
  return r;
}
template <int NL> LCPC_DEV void fe_store(u32* __restrict__ p, const Fe<NL>& a) {
  if constexpr (NL % 4 == 0) {
    uint4* q = reinterpret_cast<uint4*>(p);
#pragma unroll
    for (int i = 0; i < NL / 4; i++) q[i] = make_uint4(a.v[4 * i], a.v[4 * i + 1], a.v[4 * i + 2], a.v[4 * i + 3]);
  } else {
    uint2* q = reinterpret_cast<uint2*>(p);
#pragma unroll
    for (int i = 0; i < NL / 2; i++) q[i] = make_uint2(a.v[2 * i], a.v[2 * i + 1]);
  }
}

// ---- add / sub --------------------------------------------------------------------------------
// r = a + b mod p; 2p < 2^(32 NL) so the plain sum never carries out.
template <int NL> LCPC_DEV Fe<NL> fe_add(const Fe<NL>& a, const Fe<NL>& b) {
  Fe<NL> s, d;
  u32 c = 0;
#pragma unroll
  for (int i = 0; i < NL; i++) {
    u64 t = (u64)a.v[i] + b.v[i] + c;
    s.v[i] = (u32)t;
    c = (u32)(t >> 32);
  }
  u32 br = 0;
#pragma unroll
  for (int i = 0; i < NL; i++) {
    u64 t = (u64)s.v[i] - Mod<NL>::P[i] - br;
    d.v[i] = (u32)t;
    br = (u32)(t >> 63);
  }
#pragma unroll
  for (int i = 0; i < NL; i++) s.v[i] = br ? s.v[i] : d.v[i];
  return s;
}
template <int NL> LCPC_DEV Fe<NL> fe_sub(const Fe<NL>& a, const Fe<NL>& b) {
  Fe<NL> d, s;
  u32 br = 0;
#pragma unroll
  for (int i = 0; i < NL; i++) {
    u64 t = (u64)a.v[i] - b.v[i] - br;
    d.v[i] = (u32)t;
    br = (u32)(t >> 63);
  }
  u32 c = 0;
#pragma unroll
  for (int i = 0; i < NL; i++) {
    u64 t = (u64)d.v[i] + Mod<NL>::P[i] + c;
    s.v[i] = (u32)t;
    c = (u32)(t >> 32);
  }
#pragma unroll
  for (int i = 0; i < NL; i++) d.v[i] = br ? s.v[i] : d.v[i];
  return d;
}
// conditional final subtraction: t (NL limbs + top word) in [0, 2p) -> [0, p)
template <int NL> LCPC_DEV Fe<NL> fe_reduce_once(const u32* t, u32 top) {
  Fe<NL> d, r;
  u32 br = 0;
#pragma unroll
  for (int i = 0; i < NL; i++) {
    u64 x = (u64)t[i] - Mod<NL>::P[i] - br;
    d.v[i] = (u32)x;
    br = (u32)(x >> 63);
  }
  const bool ge = (top != 0) | (br == 0);
#pragma unroll
  for (int i = 0; i < NL; i++) r.v[i] = ge ? d.v[i] : t[i];
  return r;
}

// ---- Montgomery multiplication ----------------------------------------------------------------
// r = a*b*R^-1 mod p, fully reduced.  CIOS over 32-bit limbs, one v_mad_u64_u32 per limb product.
template <int NL> LCPC_DEV Fe<NL> fe_mul(const Fe<NL>& a, const Fe<NL>& b) {
  u32 t[NL + 2];
#pragma unroll
  for (int i = 0; i < NL + 2; i++) t[i] = 0;
#pragma unroll
  for (int i = 0; i < NL; i++) {
    u32 c = 0;
#pragma unroll
    for (int j = 0; j < NL; j++) {
      u64 s = (u64)a.v[i] * b.v[j] + t[j] + c;
      t[j] = (u32)s;
      c = (u32)(s >> 32);
    }
    u64 s = (u64)t[NL] + c;
    t[NL] = (u32)s;
    t[NL + 1] = (u32)(s >> 32);
    const u32 m = 0u - t[0];            // -p^-1 = -1 mod 2^32
    c = (t[0] != 0) ? 1u : 0u;          // carry out of t[0] + m*p[0], p[0] = 1
#pragma unroll
    for (int j = 1; j < NL; j++) {
      s = (u64)m * Mod<NL>::P[j] + t[j] + c;
      t[j - 1] = (u32)s;
      c = (u32)(s >> 32);
    }
    s = (u64)t[NL] + c;
    t[NL - 1] = (u32)s;
    t[NL] = t[NL + 1] + (u32)(s >> 32);
  }
  return fe_reduce_once<NL>(t, t[NL]);
}

// Montgomery reduction of a single element == multiply by 1: Montgomery form -> canonical value.
// This is PrimeField::to_repr (lcpc-2d/src/lib.rs:55-57) before the little-endian byte dump.
template <int NL> LCPC_DEV Fe<NL> fe_canon(const Fe<NL>& a) {
  u32 t[NL + 1];
#pragma unroll
  for (int i = 0; i < NL; i++) t[i] = a.v[i];
  t[NL] = 0;
#pragma unroll
  for (int i = 0; i < NL; i++) {
    const u32 m = 0u - t[0];
    u32 c = (t[0] != 0) ? 1u : 0u;
#pragma unroll
    for (int j = 1; j < NL; j++) {
      u64 s = (u64)m * Mod<NL>::P[j] + t[j] + c;
      t[j - 1] = (u32)s;
      c = (u32)(s >> 32);
    }
    u64 s = (u64)t[NL] + c;
    t[NL - 1] = (u32)s;
    t[NL] = (u32)(s >> 32);
  }
  return fe_reduce_once<NL>(t, t[NL]);
}


// ---- reduced-radix Montgomery multiply for Ft255 (the headline field) --------------------------
// Measured on gfx950 (profiles/r01_ubench_valu.txt): v_mad_u64_u32 issues at ~half the f32-FMA rate,
// but every carry-propagating add after it costs about as much again, and the 32-bit-limb CIOS above
// spends >2/3 of its instructions on 64-bit carry emulation.  With 9 limbs of 29 bits a whole Comba
// column (<= 9 products < 2^58 plus <= 8 reduction products) fits one 64-bit accumulator, so the
// product AND the Montgomery reduction are a pure chain of 153 v_mad_u64_u32 with no carry handling.
// p == 1 mod 2^29, hence -p^-1 == -1 mod 2^29 and the quotient digit is a negate-and-mask.
// The multiplier (a twiddle) is pre-converted on the host to the radix-2^261 Montgomery form
// w * 2^261 mod p, so  REDC_261( a*R * w*2^261 ) = (a*w)*R : data stays in ff_derive's R = 2^256 form.
struct P29 {
  static constexpr u32 M = (1u << 29) - 1;
  static constexpr u32 limb(int k) {   // k-th 29-bit limb of the Ft255 modulus
    const int b = 29 * k, w = b / 32, sh = b % 32;
    u64 lo = Mod<8>::P[w];
    u64 hi = (w + 1 < 8) ? Mod<8>::P[w + 1] : 0;
    return (u32)(((lo | (hi << 32)) >> sh) & M);
  }
};
struct Fe29 {
  u32 v[9];
};
// packed 8x32 -> 9x29 (value unchanged, limbs < 2^29)
LCPC_DEV Fe29 fe_to29(const Fe<8>& a) {
  Fe29 r;
#pragma unroll
  for (int k = 0; k < 9; k++) {
    const int b = 29 * k, w = b / 32, sh = b % 32;
    u32 x;
    if (sh == 0) x = a.v[w];
    else if (w + 1 < 8) x = __builtin_amdgcn_alignbit(a.v[w + 1], a.v[w], sh);
    else x = a.v[w] >> sh;
    r.v[k] = x & P29::M;
  }
  return r;
}
// 9x29 (limbs < 2^29, value < 2^256) -> packed 8x32
LCPC_DEV void fe_from29(u32 out[8], const u32 l[9]) {
#pragma unroll
  for (int w = 0; w < 8; w++) {
    const int b = 32 * w, k = b / 29, s = b % 29;      // word w starts at bit s of limb k
    u32 x = l[k] >> s;
    x |= l[k + 1] << (29 - s);
    if (58 - s < 32 && k + 2 < 9) x |= l[k + 2] << (58 - s);
    out[w] = x;
  }
}
// r = a * b29 * 2^-261 mod p, fully reduced, packed.  a: packed element < p; b29: 9 limbs < 2^29.
LCPC_DEV Fe<8> fe_mul_r29(const Fe<8>& a, const Fe29& b) {
  const Fe29 x = fe_to29(a);
  u32 m[9], r[9];
  u64 acc = 0;
#pragma unroll
  for (int k = 0; k < 17; k++) {
#pragma unroll
    for (int i = 0; i < 9; i++) {
      const int j = k - i;
      if (j >= 0 && j < 9) acc += (u64)x.v[i] * b.v[j];
    }
#pragma unroll
    for (int i = 0; i < 9; i++) {
      const int j = k - i;
      if (i < k && j >= 1 && j < 9) acc += (u64)m[i] * P29::limb(j);
    }
    if (k < 9) {
      m[k] = (0u - (u32)acc) & P29::M;
      acc += m[k];                    // + m_k * p_0, p_0 = 1: low 29 bits become zero
      acc >>= 29;
    } else {
      r[k - 9] = (u32)acc & P29::M;
      acc >>= 29;
    }
  }
  r[8] = (u32)acc;
  u32 t[8];
  fe_from29(t, r);
  return fe_reduce_once<8>(t, 0u);     // REDC output < 2p < 2^256
}

// ---- lazy (unreduced) accumulation: sum of products, one Montgomery reduction at the end -------
// Used by collapse_columns and the expander SpMV: acc += a*b as a plain 2NL(+1)-limb integer.
// With <= 2^32 terms of size < p^2 < 2^(64NL-2) the sum fits 2NL+1 limbs.
template <int NL> struct Wide {
  u32 v[2 * NL + 1];
};
template <int NL> LCPC_DEV Wide<NL> wide_zero() {
  Wide<NL> w;
#pragma unroll
  for (int i = 0; i < 2 * NL + 1; i++) w.v[i] = 0;
  return w;
}
template <int NL> LCPC_DEV void wide_mac(Wide<NL>& w, const Fe<NL>& a, const Fe<NL>& b) {
#pragma unroll
  for (int i = 0; i < NL; i++) {
    u32 c = 0;
#pragma unroll
    for (int j = 0; j < NL; j++) {
      u64 s = (u64)a.v[i] * b.v[j] + w.v[i + j] + c;
      w.v[i + j] = (u32)s;
      c = (u32)(s >> 32);
    }
#pragma unroll
    for (int j = i + NL; j < 2 * NL + 1; j++) {
      u64 s = (u64)w.v[j] + c;
      w.v[j] = (u32)s;
      c = (u32)(s >> 32);
    }
  }
}
// w mod p in Montgomery sense: returns w * R^-1 mod p.  The top word (bits >= 64 NL) is folded in
// first by reducing it against R^2-free arithmetic: w = lo + hi*2^(64NL) where lo < 2^(64NL);
// mont_reduce(lo) + hi * (2^(64NL) * R^-1 = 1) ... i.e. result = REDC(lo) + hi (mod p).
template <int NL> LCPC_DEV Fe<NL> wide_reduce(const Wide<NL>& w) {
  u32 t[2 * NL + 1];
#pragma unroll
  for (int i = 0; i < 2 * NL + 1; i++) t[i] = w.v[i];
  // REDC over the low 2NL limbs; carries spill into t[2NL]
#pragma unroll
  for (int i = 0; i < NL; i++) {
    const u32 m = 0u - t[i];
    u32 c = (t[i] != 0) ? 1u : 0u;
#pragma unroll
    for (int j = 1; j < NL; j++) {
      u64 s = (u64)m * Mod<NL>::P[j] + t[i + j] + c;
      t[i + j] = (u32)s;
      c = (u32)(s >> 32);
    }
#pragma unroll
    for (int j = i + NL; j < 2 * NL + 1; j++) {
      u64 s = (u64)t[j] + c;
      t[j] = (u32)s;
      c = (u32)(s >> 32);
    }
  }
  // value = t[NL .. 2NL] (NL+1 limbs), < (sum + m p)/R; bring into [0,p) by repeated subtraction
  // (top word < 2^32 terms / small: loop runs at most a few times per 2^k of terms; bounded below)
  u32 top = t[2 * NL];
  u32 x[NL];
#pragma unroll
  for (int i = 0; i < NL; i++) x[i] = t[NL + i];
  // subtract p while value >= p.  value < n_terms * p, n_terms is small in every caller (<= 2^21);
  // do it bit-serially from a shifted p to stay O(log) : here simple loop on (top:x) >= p.
  for (;;) {
    u32 d[NL];
    u32 br = 0;
#pragma unroll
    for (int i = 0; i < NL; i++) {
      u64 s = (u64)x[i] - Mod<NL>::P[i] - br;
      d[i] = (u32)s;
      br = (u32)(s >> 63);
    }
    if (top == 0 && br) break;
    top -= br;
#pragma unroll
    for (int i = 0; i < NL; i++) x[i] = d[i];
  }
  Fe<NL> r;
#pragma unroll
  for (int i = 0; i < NL; i++) r.v[i] = x[i];
  return r;
}

}  // namespace lcpc
