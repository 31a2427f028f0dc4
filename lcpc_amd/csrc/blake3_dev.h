// lcpc_amd/csrc/blake3_dev.h -- BLAKE3 compression function for gfx950, one hash state per lane.
//
// The reference is generic over digest::Digest and every test/bench instantiates blake3::Hasher
// (/root/reference/lcpc-ligero-pc/src/tests.rs:12, bench.rs:12).  Only the plain hash mode with
// 32-byte output is needed: leaf = D(0^32 || col...) (lcpc-2d/src/lib.rs:719-735) and
// parent = D(left || right) (lib.rs:770-775).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace lcpc {

enum : uint32_t { B3_CHUNK_START = 1, B3_CHUNK_END = 2, B3_PARENT = 4, B3_ROOT = 8 };

__device__ __forceinline__ uint32_t b3_rotr(uint32_t x, int n) { return __builtin_amdgcn_alignbit(x, x, n); }

#define B3_IV0 0x6A09E667u
#define B3_IV1 0xBB67AE85u
#define B3_IV2 0x3C6EF372u
#define B3_IV3 0xA54FF53Au
#define B3_IV4 0x510E527Fu
#define B3_IV5 0x9B05688Cu
#define B3_IV6 0x1F83D9ABu
#define B3_IV7 0x5BE0CD19u

__device__ __forceinline__ void b3_set_iv(uint32_t cv[8]) {
  cv[0] = B3_IV0; cv[1] = B3_IV1; cv[2] = B3_IV2; cv[3] = B3_IV3;
  cv[4] = B3_IV4; cv[5] = B3_IV5; cv[6] = B3_IV6; cv[7] = B3_IV7;
}

#define B3_G(a, b, c, d, mx, my)            \
  a = a + b + (mx); d = b3_rotr(d ^ a, 16); \
  c = c + d;        b = b3_rotr(b ^ c, 12); \
  a = a + b + (my); d = b3_rotr(d ^ a, 8);  \
  c = c + d;        b = b3_rotr(b ^ c, 7);

// One round with a compile-time message schedule: the permutation is applied to the *indices*
// (fully unrolled), so the 16 message words never move between registers.
#define B3_ROUND(m, i0, i1, i2, i3, i4, i5, i6, i7, i8, i9, i10, i11, i12, i13, i14, i15) \
  B3_G(s0, s4, s8, s12, m[i0], m[i1])   B3_G(s1, s5, s9, s13, m[i2], m[i3])               \
  B3_G(s2, s6, s10, s14, m[i4], m[i5])  B3_G(s3, s7, s11, s15, m[i6], m[i7])              \
  B3_G(s0, s5, s10, s15, m[i8], m[i9])  B3_G(s1, s6, s11, s12, m[i10], m[i11])            \
  B3_G(s2, s7, s8, s13, m[i12], m[i13]) B3_G(s3, s4, s9, s14, m[i14], m[i15])

// cv <- compress(cv, m, counter, block_len, flags)[0..8]
__device__ __forceinline__ void b3_compress(uint32_t cv[8], const uint32_t m[16], uint32_t counter_lo,
                                            uint32_t block_len, uint32_t flags) {
  uint32_t s0 = cv[0], s1 = cv[1], s2 = cv[2], s3 = cv[3], s4 = cv[4], s5 = cv[5], s6 = cv[6], s7 = cv[7];
  uint32_t s8 = B3_IV0, s9 = B3_IV1, s10 = B3_IV2, s11 = B3_IV3;
  uint32_t s12 = counter_lo, s13 = 0u, s14 = block_len, s15 = flags;
  B3_ROUND(m, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15)
  B3_ROUND(m, 2, 6, 3, 10, 7, 0, 4, 13, 1, 11, 12, 5, 9, 14, 15, 8)
  B3_ROUND(m, 3, 4, 10, 12, 13, 2, 7, 14, 6, 5, 9, 0, 11, 15, 8, 1)
  B3_ROUND(m, 10, 7, 12, 9, 14, 3, 13, 15, 4, 0, 11, 2, 5, 8, 1, 6)
  B3_ROUND(m, 12, 13, 9, 11, 15, 10, 14, 8, 7, 2, 5, 3, 0, 1, 6, 4)
  B3_ROUND(m, 9, 14, 11, 5, 8, 12, 15, 1, 13, 3, 0, 10, 2, 6, 4, 7)
  B3_ROUND(m, 11, 15, 5, 0, 1, 9, 8, 6, 14, 10, 2, 12, 3, 4, 7, 13)
  cv[0] = s0 ^ s8;  cv[1] = s1 ^ s9;  cv[2] = s2 ^ s10; cv[3] = s3 ^ s11;
  cv[4] = s4 ^ s12; cv[5] = s5 ^ s13; cv[6] = s6 ^ s14; cv[7] = s7 ^ s15;
}

// parent node of the BLAKE3 tree / D(left||right) of the Merkle tree share this shape
__device__ __forceinline__ void b3_hash64(uint32_t out[8], const uint32_t l[8], const uint32_t r[8], uint32_t flags) {
  uint32_t m[16];
#pragma unroll
  for (int i = 0; i < 8; i++) { m[i] = l[i]; m[8 + i] = r[i]; }
  b3_set_iv(out);
  b3_compress(out, m, 0u, 64u, flags);
}

// ---- one 64-byte compression spread over the four lanes of a quad ---------------------------------------------------------
// Lane q (= lane id & 3) owns column q of the 4 x 4 state, as the SIMD implementations of BLAKE3 keep one row per vector
// register: a column step is one G per lane, the diagonal step the same after rotating rows b, c, d by 1, 2, 3 lanes (DPP
// quad_perm), 42 instructions per round and lane instead of 96.  For the upper levels of a Merkle tree, where a level has
// fewer nodes than the workgroup has lanes and each level waits for the one below, this cuts the latency of a level from
// ~700 dependent instructions to ~300.  Every lane passes the same 16 message words.
template <int K> __device__ __forceinline__ uint32_t b3_qrot(uint32_t x) {      // lane l <- lane (l + K) & 3 of its quad
  constexpr int ctrl = K == 1 ? 0x39 : (K == 2 ? 0x4E : 0x93);
  return (uint32_t)__builtin_amdgcn_mov_dpp((int)x, ctrl, 0xf, 0xf, true);
}
__device__ __forceinline__ uint32_t b3_sel4(uint32_t q, uint32_t x0, uint32_t x1, uint32_t x2, uint32_t x3) {
  uint32_t r = q == 1 ? x1 : x0;
  r = q == 2 ? x2 : r;
  return q == 3 ? x3 : r;
}
#define B3_QROUND(m, i0, i1, i2, i3, i4, i5, i6, i7, i8, i9, i10, i11, i12, i13, i14, i15)                   \
  mx = b3_sel4(q, m[i0], m[i2], m[i4], m[i6]); my = b3_sel4(q, m[i1], m[i3], m[i5], m[i7]);                  \
  B3_G(a, b, c, d, mx, my)                                                                                   \
  b = b3_qrot<1>(b); c = b3_qrot<2>(c); d = b3_qrot<3>(d);                                                   \
  mx = b3_sel4(q, m[i8], m[i10], m[i12], m[i14]); my = b3_sel4(q, m[i9], m[i11], m[i13], m[i15]);            \
  B3_G(a, b, c, d, mx, my)                                                                                   \
  b = b3_qrot<3>(b); c = b3_qrot<2>(c); d = b3_qrot<1>(d);

// cv <- compress(cv, m, counter, block_len, flags) over a quad: lane q holds words q (cv_lo) and 4 + q (cv_hi) of the
// chaining value, all four lanes pass the same message block
__device__ __forceinline__ void b3_compress_quad(uint32_t q, uint32_t& cv_lo, uint32_t& cv_hi, const uint32_t m[16], uint32_t counter_lo,
                                                 uint32_t block_len, uint32_t flags) {
  uint32_t a = cv_lo, b = cv_hi;
  uint32_t c = b3_sel4(q, B3_IV0, B3_IV1, B3_IV2, B3_IV3);
  uint32_t d = b3_sel4(q, counter_lo, 0u, block_len, flags);
  uint32_t mx, my;
  B3_QROUND(m, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15)
  B3_QROUND(m, 2, 6, 3, 10, 7, 0, 4, 13, 1, 11, 12, 5, 9, 14, 15, 8)
  B3_QROUND(m, 3, 4, 10, 12, 13, 2, 7, 14, 6, 5, 9, 0, 11, 15, 8, 1)
  B3_QROUND(m, 10, 7, 12, 9, 14, 3, 13, 15, 4, 0, 11, 2, 5, 8, 1, 6)
  B3_QROUND(m, 12, 13, 9, 11, 15, 10, 14, 8, 7, 2, 5, 3, 0, 1, 6, 4)
  B3_QROUND(m, 9, 14, 11, 5, 8, 12, 15, 1, 13, 3, 0, 10, 2, 6, 4, 7)
  B3_QROUND(m, 11, 15, 5, 0, 1, 9, 8, 6, 14, 10, 2, 12, 3, 4, 7, 13)
  cv_lo = a ^ c;
  cv_hi = b ^ d;
}

// D(left || right) with the IV as chaining value (Merkle parent, lcpc-2d/src/lib.rs:770-775; BLAKE3 tree parent): all four
// lanes of the quad call it with the same l, r, flags; lane q receives words q (out_lo) and 4 + q (out_hi) of the digest
__device__ __forceinline__ void b3_hash64_quad(uint32_t q, uint32_t& out_lo, uint32_t& out_hi, const uint32_t l[8], const uint32_t r[8],
                                               uint32_t flags) {
  uint32_t m[16];
#pragma unroll
  for (int i = 0; i < 8; i++) { m[i] = l[i]; m[8 + i] = r[i]; }
  uint32_t a = b3_sel4(q, B3_IV0, B3_IV1, B3_IV2, B3_IV3);
  uint32_t b = b3_sel4(q, B3_IV4, B3_IV5, B3_IV6, B3_IV7);
  uint32_t c = a;
  uint32_t d = b3_sel4(q, 0u, 0u, 64u, flags);               // counter_lo, counter_hi, block_len, flags
  uint32_t mx, my;
  B3_QROUND(m, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15)
  B3_QROUND(m, 2, 6, 3, 10, 7, 0, 4, 13, 1, 11, 12, 5, 9, 14, 15, 8)
  B3_QROUND(m, 3, 4, 10, 12, 13, 2, 7, 14, 6, 5, 9, 0, 11, 15, 8, 1)
  B3_QROUND(m, 10, 7, 12, 9, 14, 3, 13, 15, 4, 0, 11, 2, 5, 8, 1, 6)
  B3_QROUND(m, 12, 13, 9, 11, 15, 10, 14, 8, 7, 2, 5, 3, 0, 1, 6, 4)
  B3_QROUND(m, 9, 14, 11, 5, 8, 12, 15, 1, 13, 3, 0, 10, 2, 6, 4, 7)
  B3_QROUND(m, 11, 15, 5, 0, 1, 9, 8, 6, 14, 10, 2, 12, 3, 4, 7, 13)
  out_lo = a ^ c;
  out_hi = b ^ d;
}

}  // namespace lcpc
