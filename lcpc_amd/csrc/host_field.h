// lcpc_amd/csrc/host_field.h -- host-side field descriptors and (small-volume) arithmetic.
//
// Host arithmetic is used only for one-off set-up (twiddle tables, expander-matrix generation),
// for the serial Fiat-Shamir glue of prove/verify and for the verifier's O(n_col_opens * n_rows)
// column checks -- never for the commit hot path, which runs on the GPU.
// Representation = ff_derive's [3P]: L little-endian u64 limbs, Montgomery form, R = 2^(64 L)
// (/root/reference/lcpc-test-fields/src/lib.rs:13-59).
#pragma once
#include <stdint.h>
#include <string.h>

namespace lcpc {

typedef unsigned __int128 u128;
constexpr int MAXL = 4;

struct FieldDesc {
  int id, L;
  uint64_t p[MAXL], r[MAXL], r2[MAXL], rou[MAXL];   // modulus, R, R^2, ROOT_OF_UNITY (Montgomery)
  uint64_t inv;                                      // -p^-1 mod 2^64
  uint64_t gen;                                      // multiplicative generator (PrimeFieldGenerator)
  unsigned S, num_bits;                              // 2-adicity, NUM_BITS
  uint64_t top_mask;                                 // u64::MAX >> REPR_SHAVE_BITS
  unsigned flog2() const { return num_bits - 1; }    // SizedField::FLOG2 (lcpc-2d/src/lib.rs:68-71)
};

const FieldDesc* field_desc(int id);                 // nullptr if id is not a known field

inline bool h_ge_p(const FieldDesc& f, const uint64_t* a) {
  for (int i = f.L - 1; i >= 0; i--) {
    if (a[i] > f.p[i]) return true;
    if (a[i] < f.p[i]) return false;
  }
  return true;
}
inline void h_sub_p(const FieldDesc& f, uint64_t* a) {
  uint64_t br = 0;
  for (int i = 0; i < f.L; i++) {
    u128 d = (u128)a[i] - f.p[i] - br;
    a[i] = (uint64_t)d;
    br = (uint64_t)(d >> 64) & 1;
  }
}
inline void h_add(const FieldDesc& f, uint64_t* o, const uint64_t* a, const uint64_t* b) {
  uint64_t t[MAXL], c = 0;
  for (int i = 0; i < f.L; i++) {
    u128 s = (u128)a[i] + b[i] + c;
    t[i] = (uint64_t)s;
    c = (uint64_t)(s >> 64);
  }
  if (h_ge_p(f, t)) h_sub_p(f, t);
  memcpy(o, t, 8 * f.L);
}
inline void h_sub(const FieldDesc& f, uint64_t* o, const uint64_t* a, const uint64_t* b) {
  uint64_t t[MAXL], br = 0;
  for (int i = 0; i < f.L; i++) {
    u128 d = (u128)a[i] - b[i] - br;
    t[i] = (uint64_t)d;
    br = (uint64_t)(d >> 64) & 1;
  }
  if (br) {
    uint64_t c = 0;
    for (int i = 0; i < f.L; i++) {
      u128 s = (u128)t[i] + f.p[i] + c;
      t[i] = (uint64_t)s;
      c = (uint64_t)(s >> 64);
    }
  }
  memcpy(o, t, 8 * f.L);
}
// separated operand scanning: full 2L-limb product, then L rounds of Montgomery reduction
inline void h_mul(const FieldDesc& f, uint64_t* o, const uint64_t* a, const uint64_t* b) {
  const int L = f.L;
  uint64_t t[2 * MAXL + 1] = {0};
  for (int i = 0; i < L; i++) {
    uint64_t c = 0;
    for (int j = 0; j < L; j++) {
      u128 s = (u128)a[i] * b[j] + t[i + j] + c;
      t[i + j] = (uint64_t)s;
      c = (uint64_t)(s >> 64);
    }
    t[i + L] = c;
  }
  uint64_t carry2 = 0;
  for (int i = 0; i < L; i++) {
    const uint64_t m = t[i] * f.inv;
    uint64_t c = 0;
    for (int j = 0; j < L; j++) {
      u128 s = (u128)m * f.p[j] + t[i + j] + c;
      t[i + j] = (uint64_t)s;
      c = (uint64_t)(s >> 64);
    }
    u128 s = (u128)t[i + L] + c + carry2;
    t[i + L] = (uint64_t)s;
    carry2 = (uint64_t)(s >> 64);
  }
  uint64_t r[MAXL];
  for (int i = 0; i < L; i++) r[i] = t[L + i];
  if (carry2 || h_ge_p(f, r)) h_sub_p(f, r);
  memcpy(o, r, 8 * L);
}
inline void h_canon(const FieldDesc& f, uint64_t* o, const uint64_t* a) {   // PrimeField::to_repr limbs
  uint64_t one[MAXL] = {1, 0, 0, 0};
  h_mul(f, o, a, one);
}
inline bool h_is_zero(const FieldDesc& f, const uint64_t* a) {
  uint64_t x = 0;
  for (int i = 0; i < f.L; i++) x |= a[i];
  return x == 0;
}
inline bool h_eq(const FieldDesc& f, const uint64_t* a, const uint64_t* b) {
  uint64_t x = 0;
  for (int i = 0; i < f.L; i++) x |= a[i] ^ b[i];
  return x == 0;
}

}  // namespace lcpc
