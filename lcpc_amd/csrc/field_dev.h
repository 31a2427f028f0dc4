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

// Wave priority around the memory phases of the VALU-bound kernels (the row NTTs, the column hash): a wave that is about to issue its
// few loads / LDS reads and then wait for them (tile load, the reads and twiddle loads at the top of a round, the store phase) takes
// priority 1, a wave inside its multiplier / compression chains priority 0 -- so the memory instructions of one wave are not queued
// behind hundreds of arithmetic instructions of its three neighbours on the SIMD and its latency starts to run at once.  Measured,
// same box, interleaved: headline commit 9.57 -> 9.32 ms (levels 1, 2, 3 alike).  Placement only; results are unaffected.
__device__ __forceinline__ void mem_phase(bool on) {
#ifdef __HIP_DEVICE_COMPILE__
  if (on) __builtin_amdgcn_s_setprio(1);
  else __builtin_amdgcn_s_setprio(0);
#else
  (void)on;
#endif
}

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

// ---- Ft255 add / sub as explicit VCC carry chains ---------------------------------------------
// hipcc lowers the portable u64-based add/sub above to ~90 instructions (64-bit adds + moves);
// the hardware carry chain is 8 + 8 + 8.  Two asm statements each, so that no carry flag lives
// across a statement boundary (hipcc does not model VCC inside an asm string).
#define LCPC_P8_1 "0x02a4f200"
#define LCPC_P8_2 "0x86595f30"
#define LCPC_P8_3 "0xef73c790"
#define LCPC_P8_4 "0xb9575969"
#define LCPC_P8_5 "0xfda9df04"
#define LCPC_P8_6 "0x6e4d2900"
#define LCPC_P8_7 "0x663c799b"
template <> LCPC_DEV Fe<8> fe_add<8>(const Fe<8>& a, const Fe<8>& b) {
  Fe<8> r;
  u32 d0, d1, d2, d3, d4, d5, d6, d7;
  asm("v_add_co_u32 %0, vcc, %8, %16\n\t"
      "v_addc_co_u32 %1, vcc, %9, %17, vcc\n\t"
      "v_addc_co_u32 %2, vcc, %10, %18, vcc\n\t"
      "v_addc_co_u32 %3, vcc, %11, %19, vcc\n\t"
      "v_addc_co_u32 %4, vcc, %12, %20, vcc\n\t"
      "v_addc_co_u32 %5, vcc, %13, %21, vcc\n\t"
      "v_addc_co_u32 %6, vcc, %14, %22, vcc\n\t"
      "v_addc_co_u32 %7, vcc, %15, %23, vcc"
      : "=&v"(r.v[0]), "=&v"(r.v[1]), "=&v"(r.v[2]), "=&v"(r.v[3]), "=&v"(r.v[4]), "=&v"(r.v[5]), "=&v"(r.v[6]), "=&v"(r.v[7])
      : "v"(a.v[0]), "v"(a.v[1]), "v"(a.v[2]), "v"(a.v[3]), "v"(a.v[4]), "v"(a.v[5]), "v"(a.v[6]), "v"(a.v[7]),
        "v"(b.v[0]), "v"(b.v[1]), "v"(b.v[2]), "v"(b.v[3]), "v"(b.v[4]), "v"(b.v[5]), "v"(b.v[6]), "v"(b.v[7])
      : "vcc");
  // d = r - p; keep r if that borrows (r < p)
  asm("v_subrev_co_u32 %8, vcc, 1, %0\n\t"
      "v_subbrev_co_u32 %9, vcc, %16, %1, vcc\n\t"
      "v_subbrev_co_u32 %10, vcc, %17, %2, vcc\n\t"
      "v_subbrev_co_u32 %11, vcc, %18, %3, vcc\n\t"
      "v_subbrev_co_u32 %12, vcc, %19, %4, vcc\n\t"
      "v_subbrev_co_u32 %13, vcc, %20, %5, vcc\n\t"
      "v_subbrev_co_u32 %14, vcc, %21, %6, vcc\n\t"
      "v_subbrev_co_u32 %15, vcc, %22, %7, vcc\n\t"
      "v_cndmask_b32 %0, %8, %0, vcc\n\t"
      "v_cndmask_b32 %1, %9, %1, vcc\n\t"
      "v_cndmask_b32 %2, %10, %2, vcc\n\t"
      "v_cndmask_b32 %3, %11, %3, vcc\n\t"
      "v_cndmask_b32 %4, %12, %4, vcc\n\t"
      "v_cndmask_b32 %5, %13, %5, vcc\n\t"
      "v_cndmask_b32 %6, %14, %6, vcc\n\t"
      "v_cndmask_b32 %7, %15, %7, vcc"
      : "+v"(r.v[0]), "+v"(r.v[1]), "+v"(r.v[2]), "+v"(r.v[3]), "+v"(r.v[4]), "+v"(r.v[5]), "+v"(r.v[6]), "+v"(r.v[7]),
        "=&v"(d0), "=&v"(d1), "=&v"(d2), "=&v"(d3), "=&v"(d4), "=&v"(d5), "=&v"(d6), "=&v"(d7)
      : "v"(Mod<8>::P[1]), "v"(Mod<8>::P[2]), "v"(Mod<8>::P[3]), "v"(Mod<8>::P[4]), "v"(Mod<8>::P[5]), "v"(Mod<8>::P[6]), "v"(Mod<8>::P[7])
      : "vcc");
  return r;
}
template <> LCPC_DEV Fe<8> fe_sub<8>(const Fe<8>& a, const Fe<8>& b) {
  Fe<8> r;
  u32 mask, t0, t1, t2, t3, t4, t5, t6, t7;
  asm("v_sub_co_u32 %0, vcc, %9, %17\n\t"
      "v_subb_co_u32 %1, vcc, %10, %18, vcc\n\t"
      "v_subb_co_u32 %2, vcc, %11, %19, vcc\n\t"
      "v_subb_co_u32 %3, vcc, %12, %20, vcc\n\t"
      "v_subb_co_u32 %4, vcc, %13, %21, vcc\n\t"
      "v_subb_co_u32 %5, vcc, %14, %22, vcc\n\t"
      "v_subb_co_u32 %6, vcc, %15, %23, vcc\n\t"
      "v_subb_co_u32 %7, vcc, %16, %24, vcc\n\t"
      "v_cndmask_b32 %8, 0, -1, vcc"
      : "=&v"(r.v[0]), "=&v"(r.v[1]), "=&v"(r.v[2]), "=&v"(r.v[3]), "=&v"(r.v[4]), "=&v"(r.v[5]), "=&v"(r.v[6]), "=&v"(r.v[7]),
        "=&v"(mask)
      : "v"(a.v[0]), "v"(a.v[1]), "v"(a.v[2]), "v"(a.v[3]), "v"(a.v[4]), "v"(a.v[5]), "v"(a.v[6]), "v"(a.v[7]),
        "v"(b.v[0]), "v"(b.v[1]), "v"(b.v[2]), "v"(b.v[3]), "v"(b.v[4]), "v"(b.v[5]), "v"(b.v[6]), "v"(b.v[7])
      : "vcc");
  // r += p & mask  (mask = all ones iff a < b)
  asm("v_and_b32 %8, 1, %16\n\t"
      "v_and_b32 %9, " LCPC_P8_1 ", %16\n\t"
      "v_and_b32 %10, " LCPC_P8_2 ", %16\n\t"
      "v_and_b32 %11, " LCPC_P8_3 ", %16\n\t"
      "v_and_b32 %12, " LCPC_P8_4 ", %16\n\t"
      "v_and_b32 %13, " LCPC_P8_5 ", %16\n\t"
      "v_and_b32 %14, " LCPC_P8_6 ", %16\n\t"
      "v_and_b32 %15, " LCPC_P8_7 ", %16\n\t"
      "v_add_co_u32 %0, vcc, %0, %8\n\t"
      "v_addc_co_u32 %1, vcc, %1, %9, vcc\n\t"
      "v_addc_co_u32 %2, vcc, %2, %10, vcc\n\t"
      "v_addc_co_u32 %3, vcc, %3, %11, vcc\n\t"
      "v_addc_co_u32 %4, vcc, %4, %12, vcc\n\t"
      "v_addc_co_u32 %5, vcc, %5, %13, vcc\n\t"
      "v_addc_co_u32 %6, vcc, %6, %14, vcc\n\t"
      "v_addc_co_u32 %7, vcc, %7, %15, vcc"
      : "+v"(r.v[0]), "+v"(r.v[1]), "+v"(r.v[2]), "+v"(r.v[3]), "+v"(r.v[4]), "+v"(r.v[5]), "+v"(r.v[6]), "+v"(r.v[7]),
        "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3), "=&v"(t4), "=&v"(t5), "=&v"(t6), "=&v"(t7)
      : "v"(mask)
      : "vcc");
  return r;
}
// t (8 limbs) in [0, 2p) -> [0, p): the 8-limb specialisation of fe_reduce_once with top == 0
LCPC_DEV Fe<8> fe_reduce_once8(const u32* t) {
  Fe<8> r;
#pragma unroll
  for (int i = 0; i < 8; i++) r.v[i] = t[i];
  u32 d0, d1, d2, d3, d4, d5, d6, d7;
  asm("v_subrev_co_u32 %8, vcc, 1, %0\n\t"
      "v_subbrev_co_u32 %9, vcc, %16, %1, vcc\n\t"
      "v_subbrev_co_u32 %10, vcc, %17, %2, vcc\n\t"
      "v_subbrev_co_u32 %11, vcc, %18, %3, vcc\n\t"
      "v_subbrev_co_u32 %12, vcc, %19, %4, vcc\n\t"
      "v_subbrev_co_u32 %13, vcc, %20, %5, vcc\n\t"
      "v_subbrev_co_u32 %14, vcc, %21, %6, vcc\n\t"
      "v_subbrev_co_u32 %15, vcc, %22, %7, vcc\n\t"
      "v_cndmask_b32 %0, %8, %0, vcc\n\t"
      "v_cndmask_b32 %1, %9, %1, vcc\n\t"
      "v_cndmask_b32 %2, %10, %2, vcc\n\t"
      "v_cndmask_b32 %3, %11, %3, vcc\n\t"
      "v_cndmask_b32 %4, %12, %4, vcc\n\t"
      "v_cndmask_b32 %5, %13, %5, vcc\n\t"
      "v_cndmask_b32 %6, %14, %6, vcc\n\t"
      "v_cndmask_b32 %7, %15, %7, vcc"
      : "+v"(r.v[0]), "+v"(r.v[1]), "+v"(r.v[2]), "+v"(r.v[3]), "+v"(r.v[4]), "+v"(r.v[5]), "+v"(r.v[6]), "+v"(r.v[7]),
        "=&v"(d0), "=&v"(d1), "=&v"(d2), "=&v"(d3), "=&v"(d4), "=&v"(d5), "=&v"(d6), "=&v"(d7)
      : "v"(Mod<8>::P[1]), "v"(Mod<8>::P[2]), "v"(Mod<8>::P[3]), "v"(Mod<8>::P[4]), "v"(Mod<8>::P[5]), "v"(Mod<8>::P[6]), "v"(Mod<8>::P[7])
      : "vcc");
  return r;
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
#include "field_r29_gen.h"   // r29_columns(): the 153-mad Comba/Montgomery chain as generated asm blocks
#include "field_wmul_gen.h"  // wmul_u(): x * w mod p for a WAVE-UNIFORM w given as its nine shifted multiples (scalar operands)

// r = a * b29 * 2^-261 mod p, fully reduced, packed.  a: packed element < p; b29: 9 limbs < 2^29.
LCPC_DEV Fe<8> fe_mul_r29(const Fe<8>& a, const Fe29& b) {
  const Fe29 x = fe_to29(a);
  u32 m[9], r[9];
  r29_columns(x.v, b.v, m, r);
  u32 t[8];
  fe_from29(t, r);
  return fe_reduce_once8(t);           // REDC output < 2p < 2^256
}


// =================================================================================================
// Lazy 9 x 29-bit-limb arithmetic for the Ft255 NTT (ntt_pass_l9_kernel).
//
// Measured issue costs on gfx950 (profiles/r01_ubench_valu.txt): plain v_add/v_sub/v_and/v_xor 2.4 cycles per
// wave64 instruction, but v_addc/v_subb (carry chains), v_alignbit, v_add3, 64-bit adds AND v_mad_u64_u32 all
// ~4.3-4.5.  In the packed 8x32 representation a butterfly spends ~25 % of its cycles on carry-chain add/sub and
// on packed<->29-bit conversions around the multiply.  Here an element stays in the multiplier's own format
// between stages: 9 limbs, SIGNED (two's complement) and only loosely reduced:
//     invariant I ("normalised"):  limbs 0..7 in [0, 2^29), limb 8 signed;  |value| < 4p;  value == true value (mod p).
// add and sub are 9 plain limb operations each (differences simply go negative: no bias constants, no borrows);
// the Montgomery multiply (r29_mul1s: v_mad_i64_i32 column chain with negative quotient digits) accepts limbs in
// (-2^30, 2^30), |value| < 16p, and returns a normalised value in (-1.2p, 0.2p]; a "clamp" (quotient estimate from
// the signed top limb, subtract q*p from a table) brings the only growing path (sums of sums) back into [0, 1.01p).
// Exact reduction to [0,p) and packing happen once per element per pass, at the tile store.  Bounds are stated at
// every step below and in the kernel; tests/test_gpu_edges.py::test_lazy_limb_ntt_range_stress pushes them.
// =================================================================================================
struct L9 {
  u32 v[9];       // two's complement; limb 8 (and un-normalised intermediates) may be negative
};

namespace l9 {
constexpr u32 M = P29::M;
constexpr int QOFF = 24;          // clamp table: entry i = (i - QOFF) * p, i in [0, 64)
constexpr int QBIAS = 40;         // subtracted from the top limb before the quotient estimate (keeps remainders >= 0)

LCPC_DEV L9 from_packed(const Fe<8>& a) {           // value in [0, p), normalised
  const Fe29 t = fe_to29(a);
  L9 r;
#pragma unroll
  for (int k = 0; k < 9; k++) r.v[k] = t.v[k];
  return r;
}
LCPC_DEV L9 add(const L9& a, const L9& b) {         // limb-wise, no carries
  L9 r;
#pragma unroll
  for (int k = 0; k < 9; k++) r.v[k] = a.v[k] + b.v[k];
  return r;
}
LCPC_DEV L9 sub(const L9& a, const L9& b) {         // limb-wise, limbs may go negative
  L9 r;
#pragma unroll
  for (int k = 0; k < 9; k++) r.v[k] = a.v[k] - b.v[k];
  return r;
}
// carry-propagate signed limbs (|limb| < 2^31): limbs 0..7 -> [0, 2^29), limb 8 takes what is left (signed)
LCPC_DEV void normalize(L9& a) {
#pragma unroll
  for (int k = 0; k < 8; k++) {
    a.v[k + 1] += (u32)((int32_t)a.v[k] >> 29);     // floor division: arithmetic shift
    a.v[k] &= M;
  }
}
// a: normalised, |value| < 16p  ->  normalised, value in [0, p + 2^239) (== a mod p).  qp[i] = (i - QOFF) * p as
// normalised signed limbs (12-word stride).  With t the signed top limb, V = t * 2^232 + low, 0 <= low < 2^232, and
// q = floor((t - QBIAS) / (ptop + 1)), ptop = floor(p / 2^232):  t - QBIAS = q (ptop + 1) + rem, so
// V - q p = q (2^232 - plow) + (rem + QBIAS) 2^232 + low  with plow = p mod 2^232:  >= (QBIAS - |q|) 2^232 >= 0 for
// |q| <= 17 + 1 and < (ptop + 1 + QBIAS + |q| + 1) 2^232 < p + 2^239.
LCPC_DEV void clamp(L9& a, const u32* qp) {
  constexpr u32 PTOP1 = (u32)(P29::limb(8)) + 1;                         // floor(p / 2^232) + 1 (23 bits)
  constexpr u64 MAGIC = (((u64)1 << 52) + PTOP1 - 1) / PTOP1;            // ceil(2^52 / PTOP1) < 2^30
  const u32 n = a.v[8] + (u32)(QOFF * PTOP1 - QBIAS);                    // in [0, 2^29) for |value| < 16p
  const u32 q = (u32)(((u64)n * MAGIC) >> 52);                           // exact floor(n / PTOP1) for n < 2^29
  const u32* t = qp + q * 12;
  int32_t d[9];
#pragma unroll
  for (int k = 0; k < 9; k++) d[k] = (int32_t)(a.v[k] - t[k]);
#pragma unroll
  for (int k = 0; k < 8; k++) {                                         // borrow-propagate
    d[k + 1] += d[k] >> 31;                                             // -1 if limb k went negative
    a.v[k] = (u32)d[k] & M;                                             // + 2^29 in that case
  }
  a.v[8] = (u32)d[8];
}
// ---- normalise + clamp in ONE carry pass (ntt_pass_l9s_kernel) ----------------------------------------------------
// The pure-sum output of a radix-4 butterfly, c0 = x0 + x1 + x2 + x3 of four normalised values, has limbs 0..7 in
// [0, 2^31 - 4] and a signed top limb; |value| < 16p.  clamp_q() estimates the quotient from that UN-normalised top limb
// (the carries still sitting in the lower limbs, at most 3 units of 2^232, are not in it yet), the row q*p is fetched from
// the NEGATED table nqp[i] = -(i - QOFF) * p (limb-wise; the kernel negates the l9::clamp table when it copies it to
// LDS), and clamp_apply() adds row and carries in one sweep.  With t' = t - c (c in [0, 3] the unseen carry) the
// derivation above l9::clamp gives  V - q p = q (2^232 - plow) + (rem + QBIAS + c) 2^232 + low:  still >= 0 and
// < (ptop + 1 + QBIAS + |q| + 1 + 3) 2^232 < p + 2^239.  Per limb: d = a_k - t_k + carry in (-2^29 - 1, 2^31) fits i32.
struct Row9 {
  u32 v[9];
};
LCPC_DEV u32 clamp_q(u32 top) {
  constexpr u32 PTOP1 = (u32)(P29::limb(8)) + 1;
  constexpr u64 MAGIC = (((u64)1 << 52) + PTOP1 - 1) / PTOP1;
  const u32 n = top + (u32)(QOFF * PTOP1 - QBIAS);                       // in [0, 2^29) for |value| < 16p
  return (u32)(((u64)n * MAGIC) >> 52);
}
LCPC_DEV Row9 clamp_row(const u32* nqp, u32 q) {
  const uint4 a = *reinterpret_cast<const uint4*>(nqp + q * 12), b = *reinterpret_cast<const uint4*>(nqp + q * 12 + 4);
  Row9 r;
  r.v[0] = a.x; r.v[1] = a.y; r.v[2] = a.z; r.v[3] = a.w; r.v[4] = b.x; r.v[5] = b.y; r.v[6] = b.z; r.v[7] = b.w;
  r.v[8] = nqp[q * 12 + 8];
  return r;
}
LCPC_DEV void clamp_apply(L9& a, const Row9& nt) {
  int32_t c = 0;
#pragma unroll
  for (int k = 0; k < 8; k++) {
    const int32_t d = (int32_t)(a.v[k] + nt.v[k] + (u32)c);
    a.v[k] = (u32)d & M;
    c = d >> 29;
  }
  a.v[8] = a.v[8] + nt.v[8] + (u32)c;
}
// (a * w) * 2^-261 mod p, loosely: a limbs in (-2^30, 2^30), |value| < 16p; w normalised, in [0, p) (2^261-Montgomery
// form).  Column sums stay inside i64: 9 * 2^59 + 8 * 2^58 + 2^34 < 2^63.  Result: normalised, in (-1.2p, 0.2p].
LCPC_DEV L9 mul(const L9& a, const Fe29& w) {
  L9 r;
  r29_mul1s(a.v, w.v, r.v);       // one asm statement (field_r29_gen.h)
  return r;
}
// a * w mod p for a wave-uniform w: wt = the 81 words t = 9 k + j of its shifted multiples W_j = balanced(w 2^(29 j) mod p), limb k
// (host: ctx.cpp wmul_table).  a: limbs of a difference of two normalised values (sum |limb| < 9 * 2^29).  Result: normalised,
// in (-2p, 2.7p).  119 instructions against mul()'s 188 (tools/lab, profiles/r05_ubench_wmul.jsonl: 1.6-1.8 x per second).
LCPC_DEV L9 mul_u(const L9& a, const u32* wt) {
  u32 np2[9];
#pragma unroll
  for (int j = 0; j < 9; j++) np2[j] = 0u - 2u * (u32)P29::limb(j);        // the limbs of -2p (the quotient counts units of 2p)
  L9 r;
  wmul_u(a.v, np2, wt, r.v);
  return r;
}
// exact: normalised |value| < 16p -> packed, fully reduced
LCPC_DEV Fe<8> to_packed_reduced(L9 a, const u32* qp) {
  clamp(a, qp);                   // [0, p + 2^239) < 2^256
  u32 t[8];
  fe_from29(t, a.v);
  return fe_reduce_once8(t);
}
}  // namespace l9

// Montgomery form (R = 2^256) -> canonical value for Ft255, reduction only: a * 2^-256 = REDC_261(a * 2^5).
// The "product" a << 5 needs no multiplies; the 9-step reduction is 72 v_mad_u64_u32, carry-free.
LCPC_DEV Fe<8> fe_canon_r29(const Fe<8>& a) {
  // limbs of (a << 5): bit b of the shifted value is bit b-5 of a
  u32 x[9];
#pragma unroll
  for (int k = 0; k < 9; k++) {
    const int b = 29 * k - 5;            // first source bit of limb k (negative for k = 0)
    u32 v;
    if (k == 0) v = (a.v[0] << 5);
    else {
      const int w = b / 32, sh = b % 32;
      if (sh == 0) v = a.v[w];
      else if (w + 1 < 8) v = __builtin_amdgcn_alignbit(a.v[w + 1], a.v[w], sh);
      else v = a.v[w] >> sh;
    }
    x[k] = v & P29::M;
  }
  u32 m[9], r[9];
  u64 acc = 0;
#pragma unroll
  for (int k = 0; k < 17; k++) {
    if (k < 9) acc += x[k];
#pragma unroll
    for (int i = 0; i < 9; i++) {
      const int j = k - i;
      if (i < k && j >= 1 && j < 9) acc += (u64)m[i] * P29::limb(j);
    }
    if (k < 9) {
      m[k] = (0u - (u32)acc) & P29::M;
      acc += m[k];
      acc >>= 29;
    } else {
      r[k - 9] = (u32)acc & P29::M;
      acc >>= 29;
    }
  }
  r[8] = (u32)acc;
  u32 t[8];
  fe_from29(t, r);
  return fe_reduce_once8(t);
}


// ---- carry-free lazy dot product for Ft255 (Brakedown SpMM, collapse) ---------------------------
// acc += x * v with x, v as 9 x 29-bit limbs: 81 v_mad_u64_u32 into 18 u64 columns, no carries.
// A column receives <= 9 products < 2^58 per term, so up to 7 terms fit before lazy29_normalize()
// must move the excess up; the value is Montgomery-reduced once per dot product (lazy29_reduce).
struct Lazy29 {
  u64 c[18];
};
LCPC_DEV void lazy29_zero(Lazy29& a) {
#pragma unroll
  for (int k = 0; k < 18; k++) a.c[k] = 0;
}
LCPC_DEV void lazy29_mac(Lazy29& a, const Fe29& x, const Fe29& v) {
#pragma unroll
  for (int i = 0; i < 9; i++)
#pragma unroll
    for (int j = 0; j < 9; j++) a.c[i + j] += (u64)x.v[i] * v.v[j];
}
LCPC_DEV void lazy29_normalize(Lazy29& a) {
#pragma unroll
  for (int k = 0; k < 17; k++) {
    a.c[k + 1] += a.c[k] >> 29;
    a.c[k] &= P29::M;
  }
}
// value * 2^-261 mod p, fully reduced, packed.  Requires value < 64 * p^2 (<= 64 terms of (x < p) * (v < p)),
// so that the REDC output is < 2p.  v operands must be in the 2^261-Montgomery form (like the NTT twiddles).
LCPC_DEV Fe<8> lazy29_reduce(Lazy29& a) {
  lazy29_normalize(a);
  u32 m[9], r[9];
  u64 acc = 0;
#pragma unroll
  for (int k = 0; k < 18; k++) {
    acc += a.c[k];
#pragma unroll
    for (int i = 0; i < 9; i++) {
      const int j = k - i;
      if (i < k && j >= 1 && j < 9) acc += (u64)m[i] * P29::limb(j);
    }
    if (k < 9) {
      m[k] = (0u - (u32)acc) & P29::M;
      acc += m[k];
      acc >>= 29;
    } else {
      r[k - 9] = (u32)acc & P29::M;
      acc >>= 29;
    }
  }
  u32 t[8];
  fe_from29(t, r);
  return fe_reduce_once8(t);
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
