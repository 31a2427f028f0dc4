// lcpc_amd/csrc/kernels.hip -- gfx950 (MI355X) kernels of the lcpc-2d commit / prove path.
//
//   K1  ntt_pass_kernel / ntt_pass_l9_kernel (Ft255), roots_kernel
//                               LcEncoding::encode for Ligero = fffft fft_io_pc, precomp_fft (ligero lib.rs:140, 162-164)
//   K2  transpose_to/from_t, spmm_t, sdig_rs_t (>= 24 rows); spmv, sdig_rs (fewer rows)
//                               LcEncoding::encode for Brakedown                   (brakedown encode.rs:36-110)
//   K3  leaf_chunk / leaf_finish hash_columns (+ subtree pre-merge for sharding)    (lcpc-2d lib.rs:706-745)
//   K4  merkle_subtree           merkle_tree / merkle_layer                         (lib.rs:747-785)
//   K5  collapse / collapse29 / field_sum / to_r29   collapse_columns               (lib.rs:1095-1123)
//   K6  gather_columns / gather_paths                open_column                    (lib.rs:788-825)
//
// All arithmetic is exact modular integer arithmetic, so any evaluation order gives bit-identical,
// fully-reduced results; the kernels are free to re-associate (multi-pass NTT, split sums).
#include "kernels.h"
#include "field_dev.h"
#include "ntt_l9_dev.h"
#include "field_ln.h"
#include "blake3_dev.h"
#include <algorithm>

namespace lcpc {

// limb count (runtime) -> template parameter
#define LCPC_DISPATCH_NL(nl, CALL)                 \
  switch (nl) {                                    \
    case 2: { constexpr int NLV = 2; CALL; } break; \
    case 4: { constexpr int NLV = 4; CALL; } break; \
    case 6: { constexpr int NLV = 6; CALL; } break; \
    case 8: { constexpr int NLV = 8; CALL; } break; \
    default: return hipErrorInvalidValue;          \
  }

// =================================================================================================
// K1: batched multi-pass DIF NTT (radix-4 rounds inside a pass), LDS-staged.
//
// One pass executes stages [t0, t0+s) of the n = 2^k point transform of every row.  A stage-t
// butterfly pairs x[e] and x[e + gap], gap = 2^(k-t-1), twiddle w^(2^t * (e mod gap)).  The elements
// that interact inside a pass share every index bit except bits [lb, lb+s), lb = k-t0-s, so the
// element index splits as  e = hi * 2^(lb+s) + i * 2^lb + lo.   A workgroup owns a tile of
// 2^s values of i  x  Tj = 2^log_tj consecutive values of the combined outer index (hi,lo), stages
// the tile through LDS in *memory order* (so global loads/stores are contiguous runs of
// min(Tj, 2^lb) elements -- whole 128 B lines for the first pass, the full tile for the last), and
// runs the s stages with one barrier each.  Output stays in the bit-reversed order the reference
// produces (no reorder pass); zero padding of the message is fused into the first pass's loads.
// =================================================================================================
template <int NL> struct LdsLayout {
  static constexpr int CW = (NL % 4 == 0) ? 4 : 2;      // words per LDS chunk: b128 / b64 accesses
  static constexpr int NCH = NL / CW;
};

template <int NL, int LT>
__device__ __forceinline__ Fe<NL> lds_get(const u32* lds, u32 e) {
  constexpr int CW = LdsLayout<NL>::CW, NCH = LdsLayout<NL>::NCH;
  Fe<NL> r;
#pragma unroll
  for (int c = 0; c < NCH; c++) {
    if constexpr (CW == 4) {
      uint4 t = *reinterpret_cast<const uint4*>(lds + ((size_t)c * (1u << LT) + e) * 4);
      r.v[4 * c] = t.x; r.v[4 * c + 1] = t.y; r.v[4 * c + 2] = t.z; r.v[4 * c + 3] = t.w;
    } else {
      uint2 t = *reinterpret_cast<const uint2*>(lds + ((size_t)c * (1u << LT) + e) * 2);
      r.v[2 * c] = t.x; r.v[2 * c + 1] = t.y;
    }
  }
  return r;
}
template <int NL, int LT>
__device__ __forceinline__ void lds_put(u32* lds, u32 e, const Fe<NL>& a) {
  constexpr int CW = LdsLayout<NL>::CW, NCH = LdsLayout<NL>::NCH;
#pragma unroll
  for (int c = 0; c < NCH; c++) {
    if constexpr (CW == 4)
      *reinterpret_cast<uint4*>(lds + ((size_t)c * (1u << LT) + e) * 4) =
          make_uint4(a.v[4 * c], a.v[4 * c + 1], a.v[4 * c + 2], a.v[4 * c + 3]);
    else
      *reinterpret_cast<uint2*>(lds + ((size_t)c * (1u << LT) + e) * 2) = make_uint2(a.v[2 * c], a.v[2 * c + 1]);
  }
}

// twiddle multiply: d * w^widx (Montgomery).  Ft255 uses the 29-bit-limb table (fe_mul_r29).
template <int NL> struct Tw {
  Fe<NL> w;
};
template <> struct Tw<8> {
  Fe29 w;
};
template <int NL>
__device__ __forceinline__ Tw<NL> tw_load(const NttPassArgs& a, u32 widx) {
  Tw<NL> t;
  if constexpr (NL == 8) {
    const uint4* wp = reinterpret_cast<const uint4*>(a.roots29 + (size_t)widx * 12);
    const uint4 w0 = wp[0], w1 = wp[1];
    const u32 w8 = a.roots29[(size_t)widx * 12 + 8];
    t.w.v[0] = w0.x; t.w.v[1] = w0.y; t.w.v[2] = w0.z; t.w.v[3] = w0.w;
    t.w.v[4] = w1.x; t.w.v[5] = w1.y; t.w.v[6] = w1.z; t.w.v[7] = w1.w; t.w.v[8] = w8;
  } else {
    t.w = fe_load<NL>(a.roots + (size_t)widx * NL);
  }
  return t;
}
__device__ __forceinline__ Tw<8> tw_load29(const u32* tab, u32 widx) {
  Tw<8> t;
  const uint4* wp = reinterpret_cast<const uint4*>(tab + (size_t)widx * 12);
  const uint4 w0 = wp[0], w1 = wp[1];
  const u32 w8 = tab[(size_t)widx * 12 + 8];
  t.w.v[0] = w0.x; t.w.v[1] = w0.y; t.w.v[2] = w0.z; t.w.v[3] = w0.w;
  t.w.v[4] = w1.x; t.w.v[5] = w1.y; t.w.v[6] = w1.z; t.w.v[7] = w1.w; t.w.v[8] = w8;
  return t;
}
template <int NL>
__device__ __forceinline__ Fe<NL> tw_mul(const Fe<NL>& d, const Tw<NL>& t) {
  if constexpr (NL == 8) return fe_mul_r29(d, t.w);
  else return fe_mul<NL>(d, t.w);
}

// BS threads per workgroup: 256, or 1024 when the launch has too few tiles to fill the chip (tiny commits: a round is
// then one quad per thread instead of four in sequence)
template <int NL, int LT, int BS>
__global__ void __launch_bounds__(BS) ntt_pass_kernel(NttPassArgs a) {
  extern __shared__ __attribute__((aligned(16))) u32 lds[];
  const u32 k = a.log_n, t0 = a.t0, s = a.s, ltj = a.log_tj;
  const u32 lb = k - t0 - s;                    // bits below the pass's i-field
  const u32 lbt = lb < ltj ? lb : ltj;          // lo-bits that live inside the tile
  const u32 T = 1u << (s + ltj);                // tile elements (<= 2^LT)
  const u32 tiles_per_row = 1u << (k - s - ltj);
  // XCD-aware mapping: workgroup b runs on XCD b % 8 (each XCD has a private L2).  The twiddles a tile needs
  // are the same for every row, so each XCD walks ALL rows of one tile back-to-back before moving to its next
  // tile: the tile's twiddle set (<= 2 x tile size) stays resident in that XCD's L1/L2 instead of being
  // evicted by the streamed tile data.  (Placement only affects speed, never results.)
  u64 row;
  u32 tile;
  if (tiles_per_row >= 8) {
    const u32 xcd = blockIdx.x & 7u;
    const u64 q = blockIdx.x >> 3;
    tile = (u32)(q / a.n_rows) * 8u + xcd;
    row = q % a.n_rows;
  } else {
    row = blockIdx.x / tiles_per_row;
    tile = blockIdx.x % tiles_per_row;
  }
  const u32 o0 = tile << ltj;                   // first outer index of the tile
  const u32 tid = threadIdx.x;
  const u32 lp_mask = (1u << lbt) - 1, i_mask = (1u << s) - 1;
  const u32 lo_mask = (1u << lb) - 1;

  auto gindex = [&](u32 e) -> u32 {              // LDS slot -> element index within the row (k <= 30)
    const u32 lp = e & lp_mask, i = (e >> lbt) & i_mask, hp = e >> (lbt + s);
    const u32 outer = o0 | (hp << lbt) | lp;
    return ((outer >> lb) << (lb + s)) | (i << lb) | (outer & lo_mask);
  };

  const u32* src = a.src + row * a.src_stride * NL;
  for (u32 e = tid; e < T; e += BS) {
    const u32 g = gindex(e);
    Fe<NL> v = (g < a.n_valid && row * a.src_stride + g < a.n_src_total) ? fe_load<NL>(src + (size_t)g * NL) : fe_zero<NL>();
    lds_put<NL, LT>(lds, e, v);
    // fused `coeffs` copy (lcpc-2d lib.rs:636-645): every message element is loaded exactly once, here
    if (a.copy_dst != nullptr && g < a.n_valid) fe_store<NL>(a.copy_dst + (row * a.src_stride + g) * NL, v);
  }
  __syncthreads();

  u32 u = 0;
  // ---- radix-4 rounds: local stages (u, u+1) on the four slots i0 + c*Q, c = 0..3 --------------
  for (; u + 1 < s; u += 2) {
    const u32 t = t0 + u;
    const u32 hb = s - u - 1;                    // pair bit of stage u; stage u+1 uses hb-1
    const u32 gm0 = (1u << (k - t - 1)) - 1, gm1 = gm0 >> 1;
    const bool last_two = (t + 2 == k);          // stages k-2, k-1: twiddles are 1, w^(n/4), 1
    for (u32 q = tid; q < T / 4; q += BS) {
      const u32 lp = q & lp_mask;
      const u32 j = (q >> lbt) & (i_mask >> 2);
      const u32 hp = q >> (lbt + s - 2);
      const u32 i0 = ((j >> (hb - 1)) << (hb + 1)) | (j & ((1u << (hb - 1)) - 1));
      const u32 e0 = (((hp << s) | i0) << lbt) | lp;
      const u32 dq = 1u << (hb - 1 + lbt);
      const u32 g0 = gindex(e0), g1 = gindex(e0 + dq);
      Fe<NL> x0 = lds_get<NL, LT>(lds, e0), x1 = lds_get<NL, LT>(lds, e0 + dq);
      Fe<NL> x2 = lds_get<NL, LT>(lds, e0 + 2 * dq), x3 = lds_get<NL, LT>(lds, e0 + 3 * dq);
      if (last_two) {
        // stage k-2: (x0,x2) twiddle w^0 = 1, (x1,x3) twiddle w^(n/4); stage k-1: twiddle 1 everywhere
        const Tw<NL> wq = tw_load<NL>(a, 1u << (k - 2));
        const Fe<NL> b0 = fe_add<NL>(x0, x2), b2 = fe_sub<NL>(x0, x2);
        const Fe<NL> b1 = fe_add<NL>(x1, x3), b3 = tw_mul<NL>(fe_sub<NL>(x1, x3), wq);
        lds_put<NL, LT>(lds, e0, fe_add<NL>(b0, b1));
        lds_put<NL, LT>(lds, e0 + dq, fe_sub<NL>(b0, b1));
        lds_put<NL, LT>(lds, e0 + 2 * dq, fe_add<NL>(b2, b3));
        lds_put<NL, LT>(lds, e0 + 3 * dq, fe_sub<NL>(b2, b3));
      } else {
        const Tw<NL> w0 = tw_load<NL>(a, (g0 & gm0) << t);
        const Tw<NL> w1 = tw_load<NL>(a, (g1 & gm0) << t);
        const Tw<NL> w2 = tw_load<NL>(a, (g0 & gm1) << (t + 1));
        const Fe<NL> b0 = fe_add<NL>(x0, x2), b2 = tw_mul<NL>(fe_sub<NL>(x0, x2), w0);
        const Fe<NL> b1 = fe_add<NL>(x1, x3), b3 = tw_mul<NL>(fe_sub<NL>(x1, x3), w1);
        lds_put<NL, LT>(lds, e0, fe_add<NL>(b0, b1));
        lds_put<NL, LT>(lds, e0 + dq, tw_mul<NL>(fe_sub<NL>(b0, b1), w2));
        lds_put<NL, LT>(lds, e0 + 2 * dq, fe_add<NL>(b2, b3));
        lds_put<NL, LT>(lds, e0 + 3 * dq, tw_mul<NL>(fe_sub<NL>(b2, b3), w2));
      }
    }
    __syncthreads();
  }
  // ---- radix-2 tail when s is odd --------------------------------------------------------------
  for (; u < s; u++) {
    const u32 t = t0 + u;
    const u32 hb = s - u - 1;
    const u32 gm = (1u << (k - t - 1)) - 1;
    for (u32 q = tid; q < T / 2; q += BS) {
      const u32 lp = q & lp_mask;
      const u32 j = (q >> lbt) & (i_mask >> 1);
      const u32 hp = q >> (lbt + s - 1);
      const u32 i = ((j >> hb) << (hb + 1)) | (j & ((1u << hb) - 1));
      const u32 e1 = (((hp << s) | i) << lbt) | lp;
      const u32 e2 = e1 + (1u << (hb + lbt));
      const u32 widx = (gindex(e1) & gm) << t;
      const Fe<NL> x = lds_get<NL, LT>(lds, e1), y = lds_get<NL, LT>(lds, e2);
      lds_put<NL, LT>(lds, e1, fe_add<NL>(x, y));
      if (t + 1 == k) {
        lds_put<NL, LT>(lds, e2, fe_sub<NL>(x, y));          // last stage: twiddle w^0 = 1
      } else {
        const Tw<NL> w = tw_load<NL>(a, widx);
        lds_put<NL, LT>(lds, e2, tw_mul<NL>(fe_sub<NL>(x, y), w));
      }
    }
    __syncthreads();
  }

  u32* dst = a.dst + row * a.dst_stride * NL;
  for (u32 e = tid; e < T; e += BS) fe_store<NL>(dst + (size_t)gindex(e) * NL, lds_get<NL, LT>(lds, e));
}

// -------------------------------------------------------------------------------------------------
// Ft255 variant on lazy signed 9 x 29-bit limbs (field_dev.h, namespace l9): same tiling, rounds and twiddle indexing
// as ntt_pass_kernel, but the tile lives in LDS in the multiplier's own limb format (36 B per element), add/sub are
// plain limb operations and exact reduction + packing happen once, at the tile store.  Bounds: see l9.
//
// Canonical output (a.roots29c != null; the Ligero commit): hash_columns needs to_repr(x) = x * R^-1 of every
// codeword element, one Montgomery reduction each (2^27 of them at the headline, 0.7 ms inside the hash kernel).
// The transform is linear and a twiddle multiply keeps whatever representation its input has, so the
// conversion can ride on multiplications the NTT performs anyway: after t stages exactly the elements
// [0, n / 2^t) of a row have never been multiplied ("block 0" of stage t: pure sums).  A butterfly in block 0
// multiplies its difference by the twiddle from the second table, w^i * 2^5 = (w^i * 2^261) * 2^-256, which
// converts it on the fly; every other butterfly sees canonical inputs and uses the normal table.  What is left
// in Montgomery form at the end is the prefix the trivial last stages cover (a.mont_prefix = 4 or 2 elements
// per row), reduced explicitly at the store.  comm then holds canonical values (LcCommit.coeffs, copied from the
// loads, stays in Montgomery form); the hash kernel reads them as they are.
// -------------------------------------------------------------------------------------------------
// m == ~0: -x, m == 0: x, limb-wise (per-lane choice without a select)
__device__ __forceinline__ L9 l9_neg_if(const L9& x, u32 m) {
  L9 r;
#pragma unroll
  for (int i = 0; i < 9; i++) r.v[i] = (x.v[i] ^ m) - m;
  return r;
}
template <int LT>
__global__ void __launch_bounds__(256, 4) ntt_pass_l9_kernel(NttPassArgs a) {
  constexpr int NL = 8;
  extern __shared__ __attribute__((aligned(16))) u32 lds[];
  u32* qp = lds + (size_t)Lds9<LT>::T * 9;                   // q*p table copy
  const u32 k = a.log_n, t0 = a.t0, s = a.s, ltj = a.log_tj;
  const u32 lb = k - t0 - s;
  const u32 lbt = lb < ltj ? lb : ltj;
  const u32 T = 1u << (s + ltj);
  const u32 tiles_per_row = 1u << (k - s - ltj);
  u64 row;
  u32 tile;
  if (tiles_per_row >= 8) {                                  // XCD-aware order, as in ntt_pass_kernel
    const u32 xcd = blockIdx.x & 7u;
    const u64 q = blockIdx.x >> 3;
    tile = (u32)(q / a.n_rows) * 8u + xcd;
    row = q % a.n_rows;
  } else {
    row = blockIdx.x / tiles_per_row;
    tile = blockIdx.x % tiles_per_row;
  }
  const u32 o0 = tile << ltj;
  const u32 tid = threadIdx.x;
  const u32 lp_mask = (1u << lbt) - 1, i_mask = (1u << s) - 1;
  const u32 lo_mask = (1u << lb) - 1;
  auto gindex = [&](u32 e) -> u32 {
    const u32 lp = e & lp_mask, i = (e >> lbt) & i_mask, hp = e >> (lbt + s);
    const u32 outer = o0 | (hp << lbt) | lp;
    return ((outer >> lb) << (lb + s)) | (i << lb) | (outer & lo_mask);
  };

  const bool canon = a.roots29c != nullptr;
  mem_phase(true);                                           // (field_dev.h: wave priority while a wave issues its memory instructions)
  for (u32 i = tid; i < 64 * 12; i += 256) qp[i] = a.qp29[i];
  const u32* src = a.src + row * a.src_stride * NL;
  for (u32 e = tid; e < T; e += 256) {
    const u32 g = gindex(e);
    const Fe<NL> v = (g < a.n_valid && row * a.src_stride + g < a.n_src_total) ? fe_load<NL>(src + (size_t)g * NL) : fe_zero<NL>();
    lds9_put<LT>(lds, e, l9::from_packed(v));
    if (a.copy_dst != nullptr && g < a.n_valid) fe_store<NL>(a.copy_dst + (row * a.src_stride + g) * NL, v);
  }
  __syncthreads();

  u32 u = 0;
  for (; u + 1 < s; u += 2) {
    const u32 t = t0 + u;
    const u32 hb = s - u - 1;
    const u32 gm0 = (1u << (k - t - 1)) - 1, gm1 = gm0 >> 1;
    const bool last_two = (t + 2 == k);
    // first round of a zero-padded row (rate <= 1/2): x2 = x3 = 0, the stage-0 butterflies are (x, x * w)
    const bool zero_hi = (t == 0) && !last_two && a.n_valid <= (1ull << (k - 1));
    const bool zero_3q = zero_hi && a.n_valid <= (1ull << (k - 2));
    const u32 half_mask = (1u << (k - 1)) - 1;                  // the tables hold w^i, i < n / 2
    for (u32 q = tid; q < T / 4; q += 256) {
      const u32 lp = q & lp_mask;
      const u32 j = (q >> lbt) & (i_mask >> 2);
      const u32 hp = q >> (lbt + s - 2);
      const u32 i0 = ((j >> (hb - 1)) << (hb + 1)) | (j & ((1u << (hb - 1)) - 1));
      const u32 e0 = (((hp << s) | i0) << lbt) | lp;
      const u32 dq = 1u << (hb - 1 + lbt);
      const u32 g0 = gindex(e0);
      if (zero_hi) {
        // inputs straight from the loads: value < p.  Everything is block 0 here, so with canonical output the
        // multiplies out of block 0 (w0, w1, and w2 for c1) take the converting table
        const u32* tc = canon ? a.roots29c : a.roots29;
        const u32 ex = g0 & gm0;                               // < n / 4; w0 = w^ex, w2 = w^(2 ex), w3 = w^(3 ex) = -w^(3 ex - n/2) past n/2
        const Tw<NL> w0 = tw_load29(tc, ex), w2 = tw_load29(tc, 2 * ex), w3 = tw_load29(tc, (3 * ex) & half_mask);
        const u32 ng = 3 * ex > half_mask ? ~0u : 0u;
        if (zero_3q) {           // rate <= 1/4: x1 is zero too; c0 = x0 stays where it is, three multiplies
          const L9 x0 = lds9_get<LT>(lds, e0);
          lds9_put<LT>(lds, e0 + dq, l9::mul(x0, w2.w));
          lds9_put<LT>(lds, e0 + 2 * dq, l9::mul(x0, w0.w));
          lds9_put<LT>(lds, e0 + 3 * dq, l9::mul(l9_neg_if(x0, ng), w3.w));
          continue;
        }
        const L9 x0 = lds9_get<LT>(lds, e0), x1 = lds9_get<LT>(lds, e0 + dq);
        L9 c0 = l9::add(x0, x1);                                                           // [0, 2p)
        l9::normalize(c0);
        lds9_put<LT>(lds, e0, c0);
        lds9_put<LT>(lds, e0 + dq, l9::mul(l9::sub(x0, x1), w2.w));
        const L9 tI = l9::mul_u(x1, a.wq_w);                                               // x1 I (ntt_l9s.hip: the true radix-4 form)
        lds9_put<LT>(lds, e0 + 2 * dq, l9::mul(l9::add(x0, tI), w0.w));
        lds9_put<LT>(lds, e0 + 3 * dq, l9::mul(l9_neg_if(l9::sub(x0, tI), ng), w3.w));
        continue;
      }
      const L9 x0 = lds9_get<LT>(lds, e0), x1 = lds9_get<LT>(lds, e0 + dq);
      const L9 x2 = lds9_get<LT>(lds, e0 + 2 * dq), x3 = lds9_get<LT>(lds, e0 + 3 * dq);   // I: normalised, |value| < 4p
      mem_phase(false);
      const L9 b0 = l9::add(x0, x2), b1 = l9::add(x1, x3);                                 // limbs [0, 2^30), |value| < 8p
      L9 c0 = l9::add(b0, b1);                                                             // limbs [0, 2^31), |value| < 16p
      l9::normalize(c0);
      if (last_two) {
        // stages k-2, k-1: twiddles 1, w^(n/4), 1 -- outputs go straight to the store path (normalised, |value| < 16p)
        L9 c1 = l9::sub(b0, b1);                                                           // |value| < 16p
        const L9 b2 = l9::sub(x0, x2);                                                     // limbs (-2^29, 2^29), |value| < 8p
        const L9 b3 = l9::mul_u(l9::sub(x1, x3), a.wq_w);                                  // normalised, (-2.2p, 1.5p)
        L9 c2 = l9::add(b2, b3);                                                           // |value| < 9.2p
        L9 c3 = l9::sub(b2, b3);                                                           // |value| < 9.2p
        l9::normalize(c1); l9::normalize(c2); l9::normalize(c3);
        lds9_put<LT>(lds, e0, c0);
        lds9_put<LT>(lds, e0 + dq, c1);
        lds9_put<LT>(lds, e0 + 2 * dq, c2);
        lds9_put<LT>(lds, e0 + 3 * dq, c3);
      } else {
        // block 0 of stages t, t+1 with canonical output (inputs still in Montgomery form): the three multiplies that
        // leave block 0 take the converting table; c0 stays a pure sum; c3's inputs b2, b3 are already canonical
        const bool blk0c = canon && g0 <= gm1;
        const u32* t01 = blk0c ? a.roots29c : a.roots29;
        // true radix-4 (ntt_l9s.hip): w1 = I w0, so t = (x1 - x3) I by the shifted-multiples multiply on the one constant every
        // lane shares, c2 = ((x0 - x2) + t) w0, c3 = ((x0 - x2) - t) w^(3 ex); past n/2 the table entry is the negated twiddle
        const u32 ex = (g0 & gm0) << t;
        const Tw<NL> w0 = tw_load29(t01, ex), w2 = tw_load29(t01, 2 * ex), w3 = tw_load29(t01, (3 * ex) & half_mask);
        const u32 ng = 3 * ex > half_mask ? ~0u : 0u;
        l9::clamp(c0, qp);                                                                 // [0, 1.01p)
        lds9_put<LT>(lds, e0, c0);
        lds9_put<LT>(lds, e0 + dq, l9::mul(l9::sub(b0, b1), w2.w));                        // normalised, (-1.2p, 0.2p]
        const L9 tI = l9::mul_u(l9::sub(x1, x3), a.wq_w);                                  // normalised, (-2.2p, 1.5p)
        const L9 e2 = l9::sub(x0, x2);                                                     // limbs (-2^29, 2^29), |value| < 8p
        lds9_put<LT>(lds, e0 + 2 * dq, l9::mul(l9::add(e2, tI), w0.w));
        lds9_put<LT>(lds, e0 + 3 * dq, l9::mul(l9_neg_if(l9::sub(e2, tI), ng), w3.w));
      }
      mem_phase(true);                                       // the next quad's reads, or the barrier and the next round's
    }
    __syncthreads();
  }
  for (; u < s; u++) {                                        // radix-2 tail when s is odd
    const u32 t = t0 + u;
    const u32 hb = s - u - 1;
    const u32 gm = (1u << (k - t - 1)) - 1;
    for (u32 q = tid; q < T / 2; q += 256) {
      const u32 lp = q & lp_mask;
      const u32 j = (q >> lbt) & (i_mask >> 1);
      const u32 hp = q >> (lbt + s - 1);
      const u32 i = ((j >> hb) << (hb + 1)) | (j & ((1u << hb) - 1));
      const u32 e1 = (((hp << s) | i) << lbt) | lp;
      const u32 e2 = e1 + (1u << (hb + lbt));
      const u32 g1 = gindex(e1);
      const u32 widx = (g1 & gm) << t;
      const L9 x = lds9_get<LT>(lds, e1), y = lds9_get<LT>(lds, e2);
      L9 sum = l9::add(x, y);                                                              // |value| < 8p
      l9::normalize(sum);
      l9::clamp(sum, qp);                                                                  // [0, 1.01p)
      lds9_put<LT>(lds, e1, sum);
      L9 d = l9::sub(x, y);                                                                // |value| < 8p
      if (t + 1 == k) {
        l9::normalize(d);                                                                  // last stage: twiddle 1
        lds9_put<LT>(lds, e2, d);
      } else {
        const Tw<NL> w = tw_load29((canon && g1 <= gm) ? a.roots29c : a.roots29, widx);   // block 0 of stage t converts
        lds9_put<LT>(lds, e2, l9::mul(d, w.w));
      }
    }
    __syncthreads();
  }

  u32* dst = a.dst + row * a.dst_stride * NL;
  for (u32 e = tid; e < T; e += 256) {
    const u32 g = gindex(e);
    Fe<NL> v = l9::to_packed_reduced(lds9_get<LT>(lds, e), qp);
    if (g < a.mont_prefix) v = fe_canon_r29(v);              // canonical output, final pass: the never-multiplied prefix
    fe_store<NL>(dst + (size_t)g * NL, v);
  }
}

template <int LT>
static hipError_t launch_ntt_pass_l9_t(const NttPassArgs& a, hipStream_t st) {
  const u64 tiles = ((u64)1 << (a.log_n - a.s - a.log_tj)) * a.n_rows;
  const size_t lds_bytes = (size_t)Lds9<LT>::WORDS * 4;
  // (idempotent and cheap; set on every launch rather than cached in an unsynchronised static)
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&ntt_pass_l9_kernel<LT>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                     (int)lds_bytes);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL((ntt_pass_l9_kernel<LT>), dim3((unsigned)tiles), dim3(256), lds_bytes, st, a);
  return hipGetLastError();
}

template <int NL, int LT, int BS>
static hipError_t launch_ntt_pass_bs(const NttPassArgs& a, u64 tiles, hipStream_t st) {
  const size_t lds_bytes = ((size_t)NL * 4) << LT;
  // hipFuncSetAttribute is idempotent and cheap; calling it on every launch keeps this free of unsynchronised caches
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&ntt_pass_kernel<NL, LT, BS>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL((ntt_pass_kernel<NL, LT, BS>), dim3((unsigned)tiles), dim3(BS), lds_bytes, st, a);
  return hipGetLastError();
}
template <int NL, int LT>
static hipError_t launch_ntt_pass_t(const NttPassArgs& a, hipStream_t st) {
  const u64 tiles = ((u64)1 << (a.log_n - a.s - a.log_tj)) * a.n_rows;
  // fewer tiles than CUs and at least 1024 quads per round: spend the idle SIMDs inside the workgroup
  if constexpr (LT >= 12) {
    if (tiles <= 256 && a.s + a.log_tj >= 12) return launch_ntt_pass_bs<NL, LT, 1024>(a, tiles, st);
  }
  return launch_ntt_pass_bs<NL, LT, 256>(a, tiles, st);
}
hipError_t launch_ntt_pass(int nl, int log_tile, const NttPassArgs& a, hipStream_t st) {
  if (a.s + a.log_tj > (uint32_t)log_tile) return hipErrorInvalidValue;
  if (nl == 8 && a.qp29 != nullptr) {        // Ft255: lazy 9-limb pipeline
    if (log_tile == 10) return launch_ntt_pass_l9_t<10>(a, st);
    if (log_tile == 11) return launch_ntt_pass_l9_t<11>(a, st);
    return hipErrorInvalidValue;
  }
#define NTT_CASE(NLV, LTV) if (nl == NLV && log_tile == LTV) return launch_ntt_pass_t<NLV, LTV>(a, st);
  NTT_CASE(2, 10) NTT_CASE(4, 10) NTT_CASE(6, 10) NTT_CASE(8, 10)
  NTT_CASE(2, 11) NTT_CASE(4, 11) NTT_CASE(6, 11) NTT_CASE(8, 11)
  NTT_CASE(2, 12) NTT_CASE(4, 12)
#undef NTT_CASE
  return hipErrorInvalidValue;
}

// =================================================================================================
// K3: column hashing.  leaf[c] = BLAKE3( 0^32 || to_repr(comm[0][c]) || ... || to_repr(comm[R-1][c]) ).
// The leaf message (32 + 8L*R bytes) spans several 1 KiB BLAKE3 chunks; chunk chaining values are
// independent, so the grid is (column, chunk): one lane per column (a wave reads 64 consecutive
// elements of a row = one contiguous 64*8L byte run), blockIdx.y = chunk.  This is also exactly the
// unit a row-sharded multi-GPU commit exchanges.  A second tiny kernel folds the CVs of a column with
// BLAKE3's parent rule.  Element -> canonical little-endian bytes is one Montgomery reduction.
// =================================================================================================
__device__ __forceinline__ void ld8(u32 d[8], const u32* p) {
  uint4 x = *reinterpret_cast<const uint4*>(p), y = *reinterpret_cast<const uint4*>(p + 4);
  d[0] = x.x; d[1] = x.y; d[2] = x.z; d[3] = x.w; d[4] = y.x; d[5] = y.y; d[6] = y.z; d[7] = y.w;
}
__device__ __forceinline__ void st8(u32* p, const u32 d[8]) {
  *reinterpret_cast<uint4*>(p) = make_uint4(d[0], d[1], d[2], d[3]);
  *reinterpret_cast<uint4*>(p + 4) = make_uint4(d[4], d[5], d[6], d[7]);
}
template <int NL, int PH> struct LeafRaw {
  static constexpr int NEL = (PH + 16 + NL - 1) / NL;     // elements a 16-word block touches
  Fe<NL> el[NEL];
};
// issue the global loads of one 64-byte block's elements (Montgomery form, not yet converted)
template <int NL, int PH>
__device__ __forceinline__ void leaf_load_raw(LeafRaw<NL, PH>& r, const LeafArgs& a, u64 col, int64_t row0) {
#pragma unroll
  for (int x = 0; x < LeafRaw<NL, PH>::NEL; x++) {
    const int64_t row = row0 + x;
    if (row >= 0 && (u64)row < a.n_rows_total) r.el[x] = fe_load<NL>(a.comm + ((u64)(row - a.row_base) * a.row_stride + col * a.col_stride) * NL);
    else r.el[x] = fe_zero<NL>();        // the 32-byte zero prefix (rows -1, -2, ..) and the tail past the message
  }
}
// Montgomery -> canonical little-endian words (PrimeField::to_repr), laid out as the block's 16 message words
template <int NL, int PH, bool CANON = false>
__device__ __forceinline__ void leaf_build_block(u32 m[16], const LeafRaw<NL, PH>& r) {
  Fe<NL> c[LeafRaw<NL, PH>::NEL];
#pragma unroll
  for (int x = 0; x < LeafRaw<NL, PH>::NEL; x++) {
    if constexpr (CANON) c[x] = r.el[x];                    // comm already canonical (LeafArgs::canon_in)
    else if constexpr (NL == 8) c[x] = fe_canon_r29(r.el[x]);
    else c[x] = fe_canon<NL>(r.el[x]);
  }
#pragma unroll
  for (int p = 0; p < 16; p++) m[p] = c[(PH + p) / NL].v[(PH + p) % NL];
}
template <int NL, int PH, bool CANON = false>
__device__ __forceinline__ void leaf_fill_block(u32 m[16], const LeafArgs& a, u64 col, int64_t row0) {
  LeafRaw<NL, PH> r;
  leaf_load_raw<NL, PH>(r, a, col, row0);
  leaf_build_block<NL, PH, CANON>(m, r);
}

// QUAD: four lanes per column, one compression per quad (b3_compress_quad): a small commitment has fewer (column, chunk)
// pairs than the chip has lanes, and a chunk is a chain of up to 16 dependent compressions -- ~300 instead of ~700 dependent
// instructions each.  The four lanes build the same message block (their loads coalesce to one); 64 columns per workgroup.
// Chaining value of chunk `chunk` of column `col`'s leaf message: in cv[8] (one lane per column), or spread over the quad
// (lane q: words q and 4 + q in cv_lo / cv_hi).
template <int NL, bool CANON, bool QUAD>
__device__ __forceinline__ void leaf_chunk_cv(const LeafArgs& a, u64 col, u32 chunk, u32 q, u32 cv[8], u32& cv_lo, u32& cv_hi) {
  const u64 total_len = 32 + (u64)NL * 4 * a.n_rows_total;
  const u64 chunk_off = (u64)chunk * 1024;
  const u32 chunk_len = (u32)((total_len - chunk_off) < 1024 ? (total_len - chunk_off) : 1024);
  const u32 nblocks = (chunk_len + 63) / 64;
  b3_set_iv(cv);
  cv_lo = b3_sel4(q, B3_IV0, B3_IV1, B3_IV2, B3_IV3); cv_hi = b3_sel4(q, B3_IV4, B3_IV5, B3_IV6, B3_IV7);   // QUAD: this lane's two words
  auto compress = [&](const u32* m, u32 blen, u32 flags) {
    if constexpr (QUAD) b3_compress_quad(q, cv_lo, cv_hi, m, chunk, blen, flags);
    else b3_compress(cv, m, chunk, blen, flags);
  };
  // element-word offset of block b's first word is 16*(16*chunk + b) - 8 (the zero prefix is words -8..-1)
  auto block_row0 = [&](u32 b, int& ph) -> int64_t {
    const int64_t s0 = ((int64_t)chunk * 16 + b) * 16 - 8;
    const int64_t r0 = s0 >= 0 ? s0 / NL : -((-s0 + NL - 1) / NL);
    ph = (int)(s0 - r0 * NL);
    return r0;
  };
  if constexpr (NL != 6) {
    // every block starts on an element boundary (PH == 0): software-pipeline the loads one block ahead
    // (two blocks per trip, their operands in two register sets that swap roles: no copy of the block fetched ahead)
    int ph;
    LeafRaw<NL, 0> ra, rb;
    auto one = [&](const LeafRaw<NL, 0>& r, u32 b) {
      u32 m[16];
      leaf_build_block<NL, 0, CANON>(m, r);
      const u32 rem = chunk_len - 64 * b;
      const u32 blen = rem < 64 ? rem : 64;
      u32 flags = (b == 0 ? B3_CHUNK_START : 0u);
      if (b == nblocks - 1) flags |= B3_CHUNK_END | (a.n_chunks_total == 1 ? B3_ROOT : 0u);
      compress(m, blen, flags);
    };
    leaf_load_raw<NL, 0>(ra, a, col, block_row0(0, ph));
    for (u32 b = 0; b < nblocks; b += 2) {
      if (b + 1 < nblocks) leaf_load_raw<NL, 0>(rb, a, col, block_row0(b + 1, ph));
      one(ra, b);
      if (b + 1 < nblocks) {
        if (b + 2 < nblocks) leaf_load_raw<NL, 0>(ra, a, col, block_row0(b + 2, ph));
        one(rb, b + 1);
      }
    }
  } else {
    for (u32 b = 0; b < nblocks; b++) {
      int ph;
      const int64_t row0 = block_row0(b, ph);
      u32 m[16];
      if (ph == 0) leaf_fill_block<NL, 0, CANON>(m, a, col, row0);
      else if (ph == 2) leaf_fill_block<NL, 2, CANON>(m, a, col, row0);
      else leaf_fill_block<NL, 4, CANON>(m, a, col, row0);
      const u32 rem = chunk_len - 64 * b;
      const u32 blen = rem < 64 ? rem : 64;
      u32 flags = (b == 0 ? B3_CHUNK_START : 0u);
      if (b == nblocks - 1) flags |= B3_CHUNK_END | (a.n_chunks_total == 1 ? B3_ROOT : 0u);
      compress(m, blen, flags);
    }
  }
}
template <int NL, bool CANON = false, bool QUAD = false>
__global__ void __launch_bounds__(256) leaf_chunk_kernel(LeafArgs a) {
  const u64 col = QUAD ? (u64)blockIdx.x * 64 + (threadIdx.x >> 2) : (u64)blockIdx.x * 256 + threadIdx.x;
  const u32 q = threadIdx.x & 3u;
  if (col >= a.n_cols) return;
  const u32 chunk = a.chunk_begin + blockIdx.y;
  u32 cv[8];
  u32 cv_lo, cv_hi;
  leaf_chunk_cv<NL, CANON, QUAD>(a, col, chunk, q, cv, cv_lo, cv_hi);
  u32* o = a.out + ((u64)blockIdx.y * a.n_cols + col) * 8;
  if constexpr (QUAD) {
    o[q] = cv_lo;
    o[4 + q] = cv_hi;
  } else {
    *reinterpret_cast<uint4*>(o) = make_uint4(cv[0], cv[1], cv[2], cv[3]);
    *reinterpret_cast<uint4*>(o + 4) = make_uint4(cv[4], cv[5], cv[6], cv[7]);
  }
}

// Small commitments (hash_columns + the first six levels of merkle_tree in one launch; lib.rs:706-745, 747-785): a workgroup of
// 64 quads hashes 64 columns -- the one or two chunks of a column's leaf message one after the other on its quad, folded with the
// parent rule -- and then folds its 64 leaf digests six levels up through LDS, one compression per quad, writing every level to
// its slot of `hashes`.  The launches it replaces (chunk CVs, their fold, the 512-leaf subtree kernel) are each a latency chain on
// a nearly empty chip; the tail of the tree (launch_merkle_tree_from level 6) follows.  Needs np2 == n_cols, n_cols % 64 == 0.
template <int NL, bool CANON>
__global__ void __launch_bounds__(256) leaf_tree_kernel(LeafArgs a, u32* hashes, u64 np2) {
  __shared__ u32 buf[64 * 8];
  const u32 tid = threadIdx.x, qd = tid >> 2, q = tid & 3u;
  const u64 base = (u64)blockIdx.x * 64;
  const u64 col = base + qd;
  u32 cv[8];
  u32 lo, hi;
  leaf_chunk_cv<NL, CANON, true>(a, col, 0, q, cv, lo, hi);
  if (a.n_chunks_total == 2) {
    u32 lo1, hi1, l[8], r[8];
    leaf_chunk_cv<NL, CANON, true>(a, col, 1, q, cv, lo1, hi1);
#pragma unroll
    for (int i = 0; i < 4; i++) {          // every lane of the quad needs all eight words of both chaining values
      l[i] = (u32)__shfl((int)lo, i, 4); l[4 + i] = (u32)__shfl((int)hi, i, 4);
      r[i] = (u32)__shfl((int)lo1, i, 4); r[4 + i] = (u32)__shfl((int)hi1, i, 4);
    }
    b3_hash64_quad(q, lo, hi, l, r, B3_PARENT | B3_ROOT);
  }
  {
    u32* g = hashes + col * 8;
    g[q] = lo; g[4 + q] = hi;
    buf[qd * 8 + q] = lo; buf[qd * 8 + 4 + q] = hi;
  }
  constexpr u32 FL = B3_CHUNK_START | B3_CHUNK_END | B3_ROOT;
  u64 w = np2, layer_out = np2;
  u32 n_out = 32;
#pragma unroll 1
  for (u32 j = 1; j <= 6; j++) {
    __syncthreads();
    const bool act = qd < n_out;
    u32 o_lo = 0, o_hi = 0;
    if (act) {
      u32 l[8], r[8];
      ld8(l, buf + (2 * qd) * 8);
      ld8(r, buf + (2 * qd + 1) * 8);
      b3_hash64_quad(q, o_lo, o_hi, l, r, FL);
    }
    __syncthreads();
    if (act) {
      buf[qd * 8 + q] = o_lo; buf[qd * 8 + 4 + q] = o_hi;
      u32* g = hashes + (layer_out + (base >> j) + qd) * 8;
      g[q] = o_lo; g[4 + q] = o_hi;
    }
    w >>= 1;
    layer_out += w;
    n_out >>= 1;
  }
}
bool leaf_tree_supported(const LeafArgs& a, u64 np2) {
  return a.n_chunks_total <= 2 && a.n_chunks_local == a.n_chunks_total && a.chunk_begin == 0 && np2 == a.n_cols && a.n_cols >= 128 &&
         (a.n_cols & 63) == 0 && a.n_cols * a.n_chunks_total <= 65536;
}
hipError_t launch_leaf_tree(int nl, const LeafArgs& a, u32* hashes, u64 np2, hipStream_t st) {
  if (!leaf_tree_supported(a, np2)) return hipErrorInvalidValue;
  const dim3 grid((unsigned)(a.n_cols / 64));
#define LT_CASE(NLV) case NLV: if (a.canon_in) hipLaunchKernelGGL((leaf_tree_kernel<NLV, true>), grid, dim3(256), 0, st, a, hashes, np2); \
                               else hipLaunchKernelGGL((leaf_tree_kernel<NLV, false>), grid, dim3(256), 0, st, a, hashes, np2); break;
  switch (nl) {
    LT_CASE(2) LT_CASE(4) LT_CASE(6) LT_CASE(8)
    default: return hipErrorInvalidValue;
  }
#undef LT_CASE
  return hipGetLastError();
}
// grid.y is limited to 65535: a commitment with millions of short rows (new_from_dims with a small n_per_row) has more
// leaf-message chunks than that, so the chunk range is launched in slices
template <int NL> static void launch_leaf_chunks_nl(const LeafArgs& a, hipStream_t st) {
  // few (column, chunk) pairs: four lanes per column (latency); else one lane per column (throughput)
  const bool quad = (u64)a.n_cols * a.n_chunks_local <= 65536;
  const dim3 grid((unsigned)((a.n_cols + (quad ? 63 : 255)) / (quad ? 64 : 256)), a.n_chunks_local);
  if (a.canon_in) {
    if (quad) hipLaunchKernelGGL((leaf_chunk_kernel<NL, true, true>), grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL((leaf_chunk_kernel<NL, true, false>), grid, dim3(256), 0, st, a);
  } else {
    if (quad) hipLaunchKernelGGL((leaf_chunk_kernel<NL, false, true>), grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL((leaf_chunk_kernel<NL, false, false>), grid, dim3(256), 0, st, a);
  }
}
static hipError_t launch_leaf_chunks_slice(int nl, const LeafArgs& a, hipStream_t st) {
  switch (nl) {
    case 2: launch_leaf_chunks_nl<2>(a, st); break;
    case 4: launch_leaf_chunks_nl<4>(a, st); break;
    case 6: launch_leaf_chunks_nl<6>(a, st); break;
    case 8: launch_leaf_chunks_nl<8>(a, st); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}
hipError_t launch_leaf_chunks(int nl, const LeafArgs& a, hipStream_t st) {
  if (a.n_chunks_local == 0 || a.n_cols == 0) return hipSuccess;
  constexpr u32 SLICE = 32768;
  for (u32 s0 = 0; s0 < a.n_chunks_local; s0 += SLICE) {
    LeafArgs b = a;
    b.chunk_begin = a.chunk_begin + s0;
    b.n_chunks_local = a.n_chunks_local - s0 < SLICE ? a.n_chunks_local - s0 : SLICE;
    b.out = a.out + (u64)s0 * a.n_cols * 8;
    hipError_t e = launch_leaf_chunks_slice(nl, b, st);
    if (e != hipSuccess) return e;
  }
  return hipSuccess;
}


// BLAKE3 tree over the subtree CVs ("nodes") of each column.  Node j covers 2^node_log[j] consecutive chunks
// starting at a multiple of its size (aligned power-of-two chunk groups are subtrees of the BLAKE3 tree); its CV
// lives at cvs[node_slot[j]][col].  The incremental stack algorithm, generalised to aligned subtrees: after
// adding a node of 2^l chunks, merge while bit l, l+1, ... of the running chunk count is clear.  The
// (wave-uniform) stack is kept in the slots of already-consumed nodes.  root_flag = B3_ROOT for a whole leaf
// message, 0 when this call only pre-merges a rank's chunk range into one subtree CV (row-sharded commit).
__global__ void __launch_bounds__(256) leaf_finish_kernel(u32* cvs, const u32* node_slot, const u32* node_log, u32 n_nodes,
                                                         u64 n_cols, u32* out, u32 root_flag) {
  const u64 col = (u64)blockIdx.x * 256 + threadIdx.x;
  if (col >= n_cols) return;
  auto slot = [&](u32 j) -> u64 { return node_slot ? node_slot[j] : j; };
  u32 cv[8], left[8];
  u32 len = 0;
  u64 total = 0;
  for (u32 j = 0; j < n_nodes; j++) {
    ld8(cv, cvs + (slot(j) * n_cols + col) * 8);
    if (j == n_nodes - 1) break;
    const u32 l = node_log ? node_log[j] : 0;
    total += (u64)1 << l;
    u64 t = total >> l;
    while ((t & 1) == 0) {
      --len;
      ld8(left, cvs + (slot(len) * n_cols + col) * 8);
      u32 o[8];
      b3_hash64(o, left, cv, B3_PARENT);
#pragma unroll
      for (int i = 0; i < 8; i++) cv[i] = o[i];
      t >>= 1;
    }
    st8(cvs + (slot(len) * n_cols + col) * 8, cv);
    ++len;
  }
  while (len > 0) {
    --len;
    ld8(left, cvs + (slot(len) * n_cols + col) * 8);
    u32 o[8];
    b3_hash64(o, left, cv, B3_PARENT | (len == 0 ? root_flag : 0u));
#pragma unroll
    for (int i = 0; i < 8; i++) cv[i] = o[i];
  }
  st8(out + col * 8, cv);
}
hipError_t launch_leaf_finish(u32* cvs, u32 n_chunks, u64 n_cols, u32* digests, hipStream_t st) {
  hipLaunchKernelGGL(leaf_finish_kernel, dim3((unsigned)((n_cols + 255) / 256)), dim3(256), 0, st, cvs, (const u32*)nullptr,
                     (const u32*)nullptr, n_chunks, n_cols, digests, (u32)B3_ROOT);
  return hipGetLastError();
}
hipError_t launch_leaf_finish_nodes(u32* cvs, const u32* node_slot, const u32* node_log, u32 n_nodes, u64 n_cols, u32* out,
                                    bool root, hipStream_t st) {
  hipLaunchKernelGGL(leaf_finish_kernel, dim3((unsigned)((n_cols + 255) / 256)), dim3(256), 0, st, cvs, node_slot, node_log, n_nodes,
                     n_cols, out, root ? (u32)B3_ROOT : 0u);
  return hipGetLastError();
}

// =================================================================================================
// K4: Merkle tree.  parent = D(left || right): a 64-byte message = one compression with
// CHUNK_START|CHUNK_END|ROOT.  merkle_subtree_kernel folds 2*256 = 512 nodes -> 1 inside one
// workgroup (9 layers through LDS) and writes every intermediate layer to its slot of the
// reference's flat `hashes` array (lib.rs:656-666, 747-760).
// =================================================================================================
// layers: `width` nodes at `hashes + in_off*8`; each WG reduces SUB = 2^lsub consecutive nodes down
// `lsub` layers.  Layer j (1-based) of the subtree lands at hashes[layer_off_j + wg * (SUB >> j) + ...].
template <u32 BS>
__global__ void __launch_bounds__(BS) merkle_subtree_kernel(u32* hashes, u64 in_off, u64 width, u32 lsub, u32* root_out) {
  __shared__ u32 buf[BS * 8];
  constexpr u32 NQ = BS / 4;                    // quads per workgroup
  const u32 tid = threadIdx.x;
  const u64 sub = (u64)1 << lsub;               // nodes consumed per WG (<= 2 BS)
  const u64 base = (u64)blockIdx.x * sub;
  constexpr u32 FL = B3_CHUNK_START | B3_CHUNK_END | B3_ROOT;
  u32 l[8], r[8], o[8];
  u64 layer_in = in_off, w = width;
  u64 layer_out = in_off + w;
  u32 n_out = (u32)(sub / 2);
  // A level with at most BS / 4 parents runs one compression per QUAD of lanes (b3_hash64_quad: ~300 dependent instructions
  // instead of ~700): from there on a level is shorter than the workgroup and only waits for the one below it.
  const u32 qd = tid >> 2, q = tid & 3u;
  u32 o_lo = 0, o_hi = 0;
  // first layer: read from global
  if (n_out > NQ) {
    if (tid < n_out) {
      ld8(l, hashes + (layer_in + base + 2 * tid) * 8);
      ld8(r, hashes + (layer_in + base + 2 * tid + 1) * 8);
      b3_hash64(o, l, r, FL);
      st8(hashes + (layer_out + (base >> 1) + tid) * 8, o);
      st8(buf + tid * 8, o);
    }
  } else if (qd < n_out) {
    ld8(l, hashes + (layer_in + base + 2 * qd) * 8);
    ld8(r, hashes + (layer_in + base + 2 * qd + 1) * 8);
    b3_hash64_quad(q, o_lo, o_hi, l, r, FL);
    u32* g = hashes + (layer_out + (base >> 1) + qd) * 8;
    g[q] = o_lo; g[4 + q] = o_hi;
    buf[qd * 8 + q] = o_lo; buf[qd * 8 + 4 + q] = o_hi;
  }
  for (u32 j = 2; j <= lsub; j++) {
    __syncthreads();
    layer_in = layer_out;
    w >>= 1;
    layer_out = layer_in + w;
    n_out >>= 1;
    if (n_out > NQ) {
      const bool act = tid < n_out;
      if (act) {
        ld8(l, buf + (2 * tid) * 8);
        ld8(r, buf + (2 * tid + 1) * 8);
        b3_hash64(o, l, r, FL);
      }
      __syncthreads();
      if (act) {
        st8(buf + tid * 8, o);
        st8(hashes + (layer_out + (base >> j) + tid) * 8, o);
      }
    } else {
      const bool act = qd < n_out;
      if (act) {
        ld8(l, buf + (2 * qd) * 8);
        ld8(r, buf + (2 * qd + 1) * 8);
        b3_hash64_quad(q, o_lo, o_hi, l, r, FL);
      }
      __syncthreads();
      if (act) {
        buf[qd * 8 + q] = o_lo; buf[qd * 8 + 4 + q] = o_hi;
        u32* g = hashes + (layer_out + (base >> j) + qd) * 8;
        g[q] = o_lo; g[4 + q] = o_hi;
      }
    }
  }
  // the launch that produces the root also drops it where the host wants it (pinned, device-mapped memory of the commitment):
  // the 32-byte device-to-host copy was a blit kernel of its own (4.3 us + its launch) behind every commit that returns a root
  if (root_out != nullptr) {
    __syncthreads();
    if (tid < 8) root_out[tid] = buf[tid];      // node 0 of the last level
  }
}
hipError_t launch_merkle_tree(u32* hashes, u64 np2, hipStream_t st, u32* root_out) { return launch_merkle_tree_from(hashes, np2, 0, st, root_out); }
// ... from level `levels_done` on (the levels below it are in `hashes` already: leaf_tree_kernel)
hipError_t launch_merkle_tree_from(u32* hashes, u64 np2, u32 levels_done, hipStream_t st, u32* root_out) {
  u64 in_off = 0, width = np2;
  for (u32 j = 0; j < levels_done; j++) { in_off += width; width >>= 1; }
  while (width > 1) {
    u32 lw = 0;
    while (((u64)1 << lw) < width) lw++;
    if (lw <= 9) {
      // <= 512 nodes left: the rest of the tree in ONE workgroup of 1024 threads, one compression per quad of lanes from
      // its first level on (wider levels belong on many CUs: 2048 leaves in one workgroup took 17 us against 9 + 6)
      hipLaunchKernelGGL(merkle_subtree_kernel<1024>, dim3(1), dim3(1024), 0, st, hashes, in_off, width, lw, root_out);
      return hipGetLastError();
    }
    const u32 lsub = 9;
    const u64 nwg = width >> lsub;
    hipLaunchKernelGGL(merkle_subtree_kernel<256>, dim3((unsigned)nwg), dim3(256), 0, st, hashes, in_off, width, lsub, (u32*)nullptr);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    for (u32 j = 0; j < lsub; j++) { in_off += width; width >>= 1; }
  }
  return hipSuccess;
}

// =================================================================================================
// K5: collapse_columns: poly_t[j] = sum_r coeffs[r][j] * tensor_t[r], for up to 4 tensors in one pass
// over coeffs.  Lane = column j (coalesced row reads); the tensor entry is wave-uniform.  Products
// are accumulated unreduced (2NL+1 limbs) in batches and Montgomery-reduced once per batch.
// =================================================================================================
template <int NL, int NT>
__global__ void __launch_bounds__(256) collapse_kernel(CollapseArgs a) {
  const u64 j = a.j0 + (u64)blockIdx.x * 256 + threadIdx.x;
  if (j >= a.j1) return;
  const u32 z = blockIdx.y;
  const u64 rows_per = (a.n_rows + a.n_splits - 1) / a.n_splits;
  const u64 r0 = (u64)z * rows_per;
  const u64 r1 = (r0 + rows_per < a.n_rows) ? r0 + rows_per : a.n_rows;
  constexpr int BATCH = 8;
  Fe<NL> acc[NT];
#pragma unroll
  for (int t = 0; t < NT; t++) acc[t] = fe_zero<NL>();
  for (u64 rb = r0; rb < r1; rb += BATCH) {
    Wide<NL> w[NT];
#pragma unroll
    for (int t = 0; t < NT; t++) w[t] = wide_zero<NL>();
    const u64 re = (rb + BATCH < r1) ? rb + BATCH : r1;
    for (u64 r = rb; r < re; r++) {
      const Fe<NL> c = fe_load<NL>(a.coeffs + (r * a.n_per_row + j) * NL);
#pragma unroll
      for (int t = 0; t < NT; t++) {
        const Fe<NL> tv = fe_load<NL>(a.tensors + ((u64)t * a.n_rows + r) * NL);
        wide_mac<NL>(w[t], c, tv);
      }
    }
#pragma unroll
    for (int t = 0; t < NT; t++) acc[t] = fe_add<NL>(acc[t], wide_reduce<NL>(w[t]));
  }
#pragma unroll
  for (int t = 0; t < NT; t++) fe_store<NL>(a.out + (((u64)z * NT + t) * a.out_stride + (j - a.j0)) * NL, acc[t]);
}
// Ft255: carry-free lazy dot products (lazy29_mac); the tensor entry is wave-uniform, so its nine 29-bit limbs
// (pre-converted to the 2^261 form, a.tensors29) are scalar operands of the 81 v_mad_u64_u32 per term.
template <int NT>
__global__ void __launch_bounds__(256) collapse29_kernel(CollapseArgs a) {
  const u64 j = a.j0 + (u64)blockIdx.x * 256 + threadIdx.x;
  if (j >= a.j1) return;
  const u32 z = blockIdx.y;
  const u64 rows_per = (a.n_rows + a.n_splits - 1) / a.n_splits;
  const u64 r0 = (u64)z * rows_per;
  const u64 r1 = (r0 + rows_per < a.n_rows) ? r0 + rows_per : a.n_rows;
  Fe<8> acc[NT];
#pragma unroll
  for (int t = 0; t < NT; t++) acc[t] = fe_zero<8>();
  for (u64 rb = r0; rb < r1; rb += 60) {
    const u64 re = (rb + 60 < r1) ? rb + 60 : r1;
    Lazy29 w[NT];
#pragma unroll
    for (int t = 0; t < NT; t++) lazy29_zero(w[t]);
    u32 since = 0;
    Fe<8> c = fe_load<8>(a.coeffs + (rb * a.n_per_row + j) * 8);
    for (u64 r = rb; r < re; r++) {
      Fe<8> cn = c;
      // (field_dev.h) one tensor: the next row's load goes out ahead of this row's 81 mads (0.425 -> 0.401 ms at the headline shape);
      // with two tensors fused the same costs 10 % (0.66 -> 0.73): the longer arithmetic stretch already covers the load
      if constexpr (NT == 1) mem_phase(true);
      if (r + 1 < re) cn = fe_load<8>(a.coeffs + ((r + 1) * a.n_per_row + j) * 8);     // next row in flight
      if constexpr (NT == 1) mem_phase(false);
      const Fe29 x = fe_to29(c);
#pragma unroll
      for (int t = 0; t < NT; t++) {
        Fe29 v;
#pragma unroll
        for (int i = 0; i < 9; i++) v.v[i] = __builtin_amdgcn_readfirstlane(a.tensors29[((u64)t * a.n_rows + r) * 12 + i]);
        lazy29_mac(w[t], x, v);
      }
      if (++since == 6) {
#pragma unroll
        for (int t = 0; t < NT; t++) lazy29_normalize(w[t]);
        since = 0;
      }
      c = cn;
    }
#pragma unroll
    for (int t = 0; t < NT; t++) acc[t] = fe_add<8>(acc[t], lazy29_reduce(w[t]));
  }
#pragma unroll
  for (int t = 0; t < NT; t++) fe_store<8>(a.out + (((u64)z * NT + t) * a.out_stride + (j - a.j0)) * 8, acc[t]);
}
// tensors (Montgomery, R = 2^256) -> 9 x 29-bit limbs of t * 2^261 mod p, 12-word stride
__global__ void __launch_bounds__(256) to_r29_kernel(const u32* in, u64 n, u32* out) {
  const u64 i = (u64)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  Fe<8> t = fe_load<8>(in + i * 8);
#pragma unroll
  for (int d = 0; d < 5; d++) t = fe_add<8>(t, t);
  const Fe29 x = fe_to29(t);
#pragma unroll
  for (int k = 0; k < 9; k++) out[i * 12 + k] = x.v[k];
  out[i * 12 + 9] = out[i * 12 + 10] = out[i * 12 + 11] = 0;
}
hipError_t launch_to_r29(const u32* in, u64 n, u32* out, hipStream_t st) {
  if (!n) return hipSuccess;
  hipLaunchKernelGGL(to_r29_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, in, n, out);
  return hipGetLastError();
}

template <int NL>
static hipError_t launch_collapse_t(const CollapseArgs& a, hipStream_t st) {
  dim3 grid((unsigned)((a.j1 - a.j0 + 255) / 256), a.n_splits);
  switch (a.n_tensors) {
    case 1: hipLaunchKernelGGL((collapse_kernel<NL, 1>), grid, dim3(256), 0, st, a); break;
    case 2: hipLaunchKernelGGL((collapse_kernel<NL, 2>), grid, dim3(256), 0, st, a); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}
hipError_t launch_collapse(int nl, const CollapseArgs& a_in, hipStream_t st) {
  CollapseArgs a = a_in;
  if (a.j1 == 0) { a.j0 = 0; a.j1 = a.n_per_row; }            // the whole polynomial
  if (a.out_stride == 0) a.out_stride = a.n_per_row;
  if (a.j1 > a.n_per_row || a.j0 >= a.j1) return hipErrorInvalidValue;
  if (nl == 8 && a.tensors29 != nullptr) {
    dim3 grid((unsigned)((a.j1 - a.j0 + 255) / 256), a.n_splits);
    switch (a.n_tensors) {
      case 1: hipLaunchKernelGGL(collapse29_kernel<1>, grid, dim3(256), 0, st, a); break;
      case 2: hipLaunchKernelGGL(collapse29_kernel<2>, grid, dim3(256), 0, st, a); break;
      default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
  }
  switch (nl) {
    case 2: return launch_collapse_t<2>(a, st);
    case 4: return launch_collapse_t<4>(a, st);
    case 6: return launch_collapse_t<6>(a, st);
    case 8: return launch_collapse_t<8>(a, st);
  }
  return hipErrorInvalidValue;
}

template <int NL>
__global__ void __launch_bounds__(256) field_sum_kernel(const u32* parts, u32 n_parts, u64 n_elems, u32* out) {
  const u64 e = (u64)blockIdx.x * 256 + threadIdx.x;
  if (e >= n_elems) return;
  Fe<NL> acc = fe_load<NL>(parts + e * NL);
  for (u32 p = 1; p < n_parts; p++) acc = fe_add<NL>(acc, fe_load<NL>(parts + ((u64)p * n_elems + e) * NL));
  fe_store<NL>(out + e * NL, acc);
}
hipError_t launch_field_sum(int nl, const u32* parts, u32 n_parts, u64 n_elems, u32* out, hipStream_t st) {
  dim3 grid((unsigned)((n_elems + 255) / 256));
  switch (nl) {
    case 2: hipLaunchKernelGGL(field_sum_kernel<2>, grid, dim3(256), 0, st, parts, n_parts, n_elems, out); break;
    case 4: hipLaunchKernelGGL(field_sum_kernel<4>, grid, dim3(256), 0, st, parts, n_parts, n_elems, out); break;
    case 6: hipLaunchKernelGGL(field_sum_kernel<6>, grid, dim3(256), 0, st, parts, n_parts, n_elems, out); break;
    case 8: hipLaunchKernelGGL(field_sum_kernel<8>, grid, dim3(256), 0, st, parts, n_parts, n_elems, out); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

// =================================================================================================
// K6: open_column gathers
// =================================================================================================
template <int NL>
__global__ void __launch_bounds__(256) gather_columns_kernel(const u32* comm, u64 n_rows, u64 row_stride, u64 col_stride, const u64* cols,
                                                            u32* vals, const u32* r2) {
  const u32 k = blockIdx.y;
  const u64 c = cols[k];
  for (u64 r = (u64)blockIdx.x * 256 + threadIdx.x; r < n_rows; r += (u64)gridDim.x * 256) {
    Fe<NL> v = fe_load<NL>(comm + (r * row_stride + c * col_stride) * NL);
    if (r2 != nullptr) v = fe_mul<NL>(v, fe_load<NL>(r2));  // canonical comm -> Montgomery form
    fe_store<NL>(vals + ((u64)k * n_rows + r) * NL, v);
  }
}
template <int NL>
__global__ void __launch_bounds__(256) to_mont_kernel(const u32* in, u64 n, const u32* r2, u32* out) {
  const Fe<NL> rr = fe_load<NL>(r2);
  for (u64 i = (u64)blockIdx.x * 256 + threadIdx.x; i < n; i += (u64)gridDim.x * 256)
    fe_store<NL>(out + i * NL, fe_mul<NL>(fe_load<NL>(in + i * NL), rr));
}
template <int NL>
__global__ void __launch_bounds__(256) to_canon_kernel(const u32* in, u64 n, u32* out) {
  for (u64 i = (u64)blockIdx.x * 256 + threadIdx.x; i < n; i += (u64)gridDim.x * 256)
    fe_store<NL>(out + i * NL, fe_canon<NL>(fe_load<NL>(in + i * NL)));
}
hipError_t launch_to_mont(int nl, const u32* in, u64 n, const u32* r2, u32* out, hipStream_t st) {
  if (n == 0) return hipSuccess;
  const unsigned gx = (unsigned)std::min<u64>((n + 255) / 256, 65536);
  LCPC_DISPATCH_NL(nl, hipLaunchKernelGGL(to_mont_kernel<NLV>, dim3(gx), dim3(256), 0, st, in, n, r2, out));
  return hipGetLastError();
}
hipError_t launch_to_canon(int nl, const u32* in, u64 n, u32* out, hipStream_t st) {
  if (n == 0) return hipSuccess;
  const unsigned gx = (unsigned)std::min<u64>((n + 255) / 256, 65536);
  LCPC_DISPATCH_NL(nl, hipLaunchKernelGGL(to_canon_kernel<NLV>, dim3(gx), dim3(256), 0, st, in, n, out));
  return hipGetLastError();
}
hipError_t launch_gather_columns(int nl, const u32* comm, u64 n_rows, u64 row_stride, u64 col_stride, const u64* cols, u32 n, u32* vals,
                                 const u32* r2, hipStream_t st) {
  if (n == 0) return hipSuccess;
  unsigned gx = (unsigned)((n_rows + 255) / 256);
  if (gx > 64) gx = 64;
  constexpr u32 SLICE = 32768;                 // grid.y limit: columns in slices
  for (u32 s0 = 0; s0 < n; s0 += SLICE) {
    const u32 cnt = n - s0 < SLICE ? n - s0 : SLICE;
    dim3 grid(gx, cnt);
    const u64* c = cols + s0;
    u32* v = vals + (u64)s0 * n_rows * nl;
    switch (nl) {
      case 2: hipLaunchKernelGGL(gather_columns_kernel<2>, grid, dim3(256), 0, st, comm, n_rows, row_stride, col_stride, c, v, r2); break;
      case 4: hipLaunchKernelGGL(gather_columns_kernel<4>, grid, dim3(256), 0, st, comm, n_rows, row_stride, col_stride, c, v, r2); break;
      case 6: hipLaunchKernelGGL(gather_columns_kernel<6>, grid, dim3(256), 0, st, comm, n_rows, row_stride, col_stride, c, v, r2); break;
      case 8: hipLaunchKernelGGL(gather_columns_kernel<8>, grid, dim3(256), 0, st, comm, n_rows, row_stride, col_stride, c, v, r2); break;
      default: return hipErrorInvalidValue;
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
  }
  return hipSuccess;
}
__global__ void __launch_bounds__(256) gather_paths_kernel(const u32* hashes, u64 np2, u32 path_len, const u64* cols, u32 n,
                                                          u32* paths) {
  const u64 id = (u64)blockIdx.x * 256 + threadIdx.x;
  if (id >= (u64)n * path_len) return;
  const u32 k = (u32)(id / path_len), lvl = (u32)(id % path_len);
  u64 base = 0, w = np2;
  for (u32 i = 0; i < lvl; i++) { base += w; w >>= 1; }
  const u64 node = (cols[k] >> lvl) ^ 1;
  u32 d[8];
  ld8(d, hashes + (base + node) * 8);
  st8(paths + id * 8, d);
}
// sharded open_column: rank g's block of the all-gather holds its rows of the n opened columns as [k][rows of g]; the proof
// wants [k][all rows].  rb[0..G]: first row of every rank (rb[G] = n_rows), device-resident.  One element per thread.
__global__ void __launch_bounds__(256) assemble_columns_kernel(const u32* recv, u64 block_words, const u64* rb, u32 G, u32 n, u64 n_rows,
                                                              u32 nl, u32* out) {
  const u64 id = (u64)blockIdx.x * 256 + threadIdx.x;
  if (id >= (u64)n * n_rows) return;
  const u64 k = id / n_rows, r = id % n_rows;
  u32 g = 0;
  while (g + 1 < G && r >= rb[g + 1]) g++;
  const u64 nr_g = rb[g + 1] - rb[g];
  const u32* src = recv + (u64)g * block_words + (k * nr_g + (r - rb[g])) * nl;
  u32* dst = out + id * nl;
  for (u32 w = 0; w < nl; w += 2) *reinterpret_cast<uint2*>(dst + w) = *reinterpret_cast<const uint2*>(src + w);
}
hipError_t launch_assemble_columns(int nl, const u32* recv, u64 block_words, const u64* rb, u32 G, u32 n, u64 n_rows, u32* out, hipStream_t st) {
  const u64 tot = (u64)n * n_rows;
  if (tot == 0) return hipSuccess;
  hipLaunchKernelGGL(assemble_columns_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, recv, block_words, rb, G, n, n_rows, (u32)nl, out);
  return hipGetLastError();
}
hipError_t launch_gather_paths(const u32* hashes, u64 np2, u32 path_len, const u64* cols, u32 n, u32* paths, hipStream_t st) {
  const u64 tot = (u64)n * path_len;
  if (tot == 0) return hipSuccess;
  hipLaunchKernelGGL(gather_paths_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, hashes, np2, path_len, cols, n,
                     paths);
  return hipGetLastError();
}

// =================================================================================================
// K2: Brakedown expander code.  The reference multiplies a CSC matrix by one row at a time
// (sprs CsMat::dot, encode.rs:50-55); here the matrix is stored CSR-by-output and applied to all
// rows of the commitment in one launch: lane = output index o (consecutive lanes write consecutive
// codeword positions), blockIdx.y = matrix row of the commitment.  Products accumulate unreduced.
// =================================================================================================
// SL lanes per output: lane s of the group takes terms k0 + s, k0 + s + SL, ..; the partial sums (exact field elements) are
// added across the group with lane shuffles.  A launch of this path has a few rows only (the verifier's 1 + n_degree_tests
// single-row encodes, commitments of < 24 rows) and is bound by the latency of an output's ~45 dependent index -> gather ->
// multiply steps, not by throughput: eight lanes per output cut that chain to ~6 steps.
template <int NL, int SL>
__global__ void __launch_bounds__(256) spmv_kernel(SpmvArgs a) {
  const u64 gid = (u64)blockIdx.x * 256 + threadIdx.x;
  const u64 o = gid / SL;
  const u32 sl = (u32)(gid % SL);
  const bool live = o < a.m;                         // whole groups are live or dead together (256 % SL == 0)
  const u64 oo = live ? o : 0;
  const u64 row = blockIdx.y;
  const u32* x = a.mat + (row * a.stride + a.in_off) * NL;
  const u32 k0 = a.rowptr[oo], k1 = a.rowptr[oo + 1];
  Fe<NL> acc = fe_zero<NL>();
  bool lazy = false;
  if constexpr (NL == 8) {
    lazy = a.vals29 != nullptr && k1 - k0 <= 60 * SL;  // lazy29_reduce takes <= 60 terms: a lane sees ceil((k1 - k0) / SL)
    if (lazy) {
      // the carry-free 29-bit-limb dot product of the position-major kernels
      Lazy29 l;
      lazy29_zero(l);
      u32 since = 0;
      for (u32 k = k0 + sl; k < k1; k += SL) {
        Fe29 v;
#pragma unroll
        for (int i = 0; i < 9; i++) v.v[i] = a.vals29[(size_t)k * 12 + i];
        lazy29_mac(l, fe_to29(fe_load<NL>(x + (u64)a.colidx[k] * NL)), v);
        if (++since == 6) { lazy29_normalize(l); since = 0; }
      }
      acc = lazy29_reduce(l);
    }
  }
  if (!lazy) {
    constexpr u32 BATCH = 8;
    for (u32 kb = k0 + sl; kb < k1; kb += BATCH * SL) {
      Wide<NL> w = wide_zero<NL>();
#pragma unroll 2
      for (u32 i = 0; i < BATCH; i++) {
        const u32 k = kb + i * SL;
        if (k >= k1) break;
        wide_mac<NL>(w, fe_load<NL>(a.vals + (u64)k * NL), fe_load<NL>(x + (u64)a.colidx[k] * NL));
      }
      acc = fe_add<NL>(acc, wide_reduce<NL>(w));
    }
  }
#pragma unroll
  for (int off = SL / 2; off > 0; off >>= 1) {
    Fe<NL> other;
#pragma unroll
    for (int i = 0; i < NL; i++) other.v[i] = (u32)__shfl_xor((int)acc.v[i], off, 64);
    acc = fe_add<NL>(acc, other);
  }
  if (live && sl == 0) {
    u32* dst = a.out_alt ? a.out_alt + (row * a.out_alt_stride + o) * NL : a.mat + (row * a.stride + a.out_off + o) * NL;
    fe_store<NL>(dst, acc);
  }
}
hipError_t launch_spmv(int nl, const SpmvArgs& a, hipStream_t st) {
  if (a.m == 0 || a.n_rows == 0) return hipSuccess;
  if (a.n_rows > 65535) return hipErrorInvalidValue;        // (the host takes the position-major path from 24 rows on)
  constexpr int SL = 8;
  dim3 grid((unsigned)((a.m * SL + 255) / 256), (unsigned)a.n_rows);
  LCPC_DISPATCH_NL(nl, hipLaunchKernelGGL((spmv_kernel<NLV, SL>), grid, dim3(256), 0, st, a));
  return hipGetLastError();
}

template <int NL>
__global__ void __launch_bounds__(64) sdig_rs_kernel(const u32* in, u64 in_stride, u32 n_in, u32* mat, u64 stride, u64 out_off,
                                                    u32 n_out, const u32* r2) {
  const u32 k = threadIdx.x + blockIdx.x * 64;
  if (k >= n_out) return;
  const u64 row = blockIdx.y;
  // x = (k+1) in Montgomery form = (k+1) * R^2 * R^-1
  Fe<NL> raw = fe_zero<NL>();
  raw.v[0] = k + 1;
  const Fe<NL> x = fe_mul<NL>(raw, fe_load<NL>(r2));
  Fe<NL> r = fe_zero<NL>();
  for (u32 j = n_in; j-- > 0;) r = fe_add<NL>(fe_mul<NL>(r, x), fe_load<NL>(in + (row * in_stride + j) * NL));
  fe_store<NL>(mat + (row * stride + out_off + k) * NL, r);
}
hipError_t launch_sdig_rs(int nl, const u32* in, u64 in_stride, u32 n_in, u32* mat, u64 stride, u64 out_off, u32 n_out,
                          u64 n_rows, const u32* r2, hipStream_t st) {
  if (n_out == 0 || n_rows == 0) return hipSuccess;
  if (n_rows > 65535) return hipErrorInvalidValue;
  dim3 grid((n_out + 63) / 64, (unsigned)n_rows);
  switch (nl) {
    case 2: hipLaunchKernelGGL(sdig_rs_kernel<2>, grid, dim3(64), 0, st, in, in_stride, n_in, mat, stride, out_off, n_out, r2); break;
    case 4: hipLaunchKernelGGL(sdig_rs_kernel<4>, grid, dim3(64), 0, st, in, in_stride, n_in, mat, stride, out_off, n_out, r2); break;
    case 6: hipLaunchKernelGGL(sdig_rs_kernel<6>, grid, dim3(64), 0, st, in, in_stride, n_in, mat, stride, out_off, n_out, r2); break;
    case 8: hipLaunchKernelGGL(sdig_rs_kernel<8>, grid, dim3(64), 0, st, in, in_stride, n_in, mat, stride, out_off, n_out, r2); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}


// =================================================================================================
// K2 (fast path): Brakedown on a position-major copy T[pos][row] of the commitment rows.
// The expander matrix is the same for every row, so with lane = row the CSR entries (column index, value)
// are wave-uniform (scalar loads / SGPR multiplier operands) and every gathered operand is one contiguous
// n_rows * F byte run.  Two tiled transposes (in: message, out: whole codeword) bracket the level chain.
// =================================================================================================
template <int NL>
__global__ void __launch_bounds__(256) transpose_to_t_kernel(const u32* src, u64 src_stride, u64 n_valid, u64 n_rows, u32* t,
                                                            u64 n_src_total, u32* copy_dst, u32 canon) {
  // tile: 32 positions x 32 rows; LDS holds it row-major with a one-element pad
  __shared__ u32 tile[32 * 33 * NL];
  const u64 p0 = (u64)blockIdx.x * 32, r0 = (u64)blockIdx.y * 32;
  const u32 tx = threadIdx.x & 31, ty = threadIdx.x >> 5;       // 32 x 8
  for (u32 rr = ty; rr < 32; rr += 8) {
    const u64 r = r0 + rr, p = p0 + tx;
    if (r < n_rows && p < n_valid) {
      // flat source elements >= n_src_total read as zero (ragged last row); the padded LcCommit.coeffs copy
      // (lcpc-2d lib.rs:636-645) is written here, where every message element is loaded exactly once
      const Fe<NL> v = (r * src_stride + p < n_src_total) ? fe_load<NL>(src + (r * src_stride + p) * NL) : fe_zero<NL>();
      if (copy_dst != nullptr) fe_store<NL>(copy_dst + (r * src_stride + p) * NL, v);
      // canon: the working copy starts from canonical values x R^-1.  Every later step is linear with Montgomery-form
      // constants (dot products with the matrix values, Horner in the R-S base case), i.e. keeps the form of its input, so the
      // whole codeword comes out canonical and the column hash reads it as it is -- the one reduction per element happens
      // here, in a kernel that waits for memory, instead of in the hash kernel, which is bound by VALU issue
      Fe<NL> u = v;
      if (canon) {
        if constexpr (NL == 8) u = fe_canon_r29(v);            // 72 carry-free mads instead of the packed-limb reduction
        else u = fe_canon<NL>(v);
      }
#pragma unroll
      for (int w = 0; w < NL; w++) tile[(rr * 33 + tx) * NL + w] = u.v[w];
    }
  }
  __syncthreads();
  for (u32 pp = ty; pp < 32; pp += 8) {
    const u64 p = p0 + pp, r = r0 + tx;
    if (r < n_rows && p < n_valid) {
      Fe<NL> v;
#pragma unroll
      for (int w = 0; w < NL; w++) v.v[w] = tile[(tx * 33 + pp) * NL + w];
      fe_store<NL>(t + (p * n_rows + r) * NL, v);
    }
  }
}
template <int NL>
__global__ void __launch_bounds__(256) transpose_from_t_kernel(const u32* t, u64 n_pos, u64 n_rows, u32* dst, u64 dst_stride) {
  __shared__ u32 tile[32 * 33 * NL];
  const u64 p0 = (u64)blockIdx.x * 32, r0 = (u64)blockIdx.y * 32;
  const u32 tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (u32 pp = ty; pp < 32; pp += 8) {
    const u64 p = p0 + pp, r = r0 + tx;
    if (r < n_rows && p < n_pos) {
      const Fe<NL> v = fe_load<NL>(t + (p * n_rows + r) * NL);
#pragma unroll
      for (int w = 0; w < NL; w++) tile[(pp * 33 + tx) * NL + w] = v.v[w];
    }
  }
  __syncthreads();
  for (u32 rr = ty; rr < 32; rr += 8) {
    const u64 r = r0 + rr, p = p0 + tx;
    if (r < n_rows && p < n_pos) {
      Fe<NL> v;
#pragma unroll
      for (int w = 0; w < NL; w++) v.v[w] = tile[(tx * 33 + rr) * NL + w];
      fe_store<NL>(dst + (r * dst_stride + p) * NL, v);
    }
  }
}
hipError_t launch_transpose_to_t(int nl, const u32* src, u64 src_stride, u64 n_valid, u64 n_rows, u32* t, hipStream_t st,
                                 u64 n_src_total, u32* copy_dst, bool canon) {
  if (!n_valid || !n_rows) return hipSuccess;
  dim3 grid((unsigned)((n_valid + 31) / 32), (unsigned)((n_rows + 31) / 32));
  LCPC_DISPATCH_NL(nl, hipLaunchKernelGGL(transpose_to_t_kernel<NLV>, grid, dim3(256), 0, st, src, src_stride, n_valid, n_rows, t,
                                          n_src_total, copy_dst, canon ? 1u : 0u));
  return hipGetLastError();
}
hipError_t launch_transpose_from_t(int nl, const u32* t, u64 n_pos, u64 n_rows, u32* dst, u64 dst_stride, hipStream_t st) {
  if (!n_pos || !n_rows) return hipSuccess;
  dim3 grid((unsigned)((n_pos + 31) / 32), (unsigned)((n_rows + 31) / 32));
  LCPC_DISPATCH_NL(nl, hipLaunchKernelGGL(transpose_from_t_kernel<NLV>, grid, dim3(256), 0, st, t, n_pos, n_rows, dst, dst_stride));
  return hipGetLastError();
}

typedef u32 __attribute__((address_space(4))) ConstU32;
// one output's partial dot product over terms [k0, k1) of its CSR row (wave-uniform bounds), gathered from column rr
template <int NL>
__device__ __forceinline__ Fe<NL> spmm_t_terms(const SpmmTArgs& a, const u32* xin, size_t pstride, u32 k0, u32 k1) {
  Fe<NL> res = fe_zero<NL>();
  if constexpr (NL == 8) {
    // The matrix is read through the constant address space: its wave-uniform addresses become scalar loads (s_load
    // into SGPRs, which the multiplier takes directly) instead of vector loads + readfirstlane (-10 VALU per term).
    const ConstU32* cidx = (const ConstU32*)a.colidx;
    const ConstU32* cv29 = (const ConstU32*)a.vals29;
    for (u32 kb = k0; kb < k1; kb += 60) {         // <= 60 terms per Montgomery reduction (lazy29_reduce bound)
      const u32 ke = kb + 60 < k1 ? kb + 60 : k1;
      Lazy29 acc;
      lazy29_zero(acc);
      u32 since = 0;
      // the column index is fetched two terms ahead, the gathered operand and the matrix value one term ahead (indices
      // clamped to the chunk: the redundant scalar loads at its end stay in bounds)
      Fe<NL> x = fe_load<NL>(xin + (size_t)cidx[kb] * pstride);
      u32 cn = cidx[kb + 1 < ke ? kb + 1 : kb];
      Fe29 v;
#pragma unroll
      for (int i = 0; i < 9; i++) v.v[i] = cv29[(size_t)kb * 12 + i];
      for (u32 k = kb; k < ke; k++) {
        Fe<NL> xn = x;
        mem_phase(true);                            // (field_dev.h) the next term's gather goes out ahead of this term's 81 mads
        if (k + 1 < ke) xn = fe_load<NL>(xin + (size_t)cn * pstride);
        const u32 kn = k + 1 < ke ? k + 1 : k, kn2 = k + 2 < ke ? k + 2 : k;
        cn = cidx[kn2];
        Fe29 vn;
#pragma unroll
        for (int i = 0; i < 9; i++) vn.v[i] = cv29[(size_t)kn * 12 + i];
        mem_phase(false);
        lazy29_mac(acc, fe_to29(x), v);
        v = vn;
        if (++since == 6) { lazy29_normalize(acc); since = 0; }
        x = xn;
      }
      res = fe_add<NL>(res, lazy29_reduce(acc));
    }
  } else {
    bool done = false;
    if constexpr (NL == 4 || NL == 6) {
      if (a.vals29 != nullptr) {
        done = true;
        // Ft127 / Ft191: the same carry-free limb dot product on 5 / 7 limbs of 29 bits (field_ln.h); matrix values in the
        // R'-Montgomery limb form (8-word stride), read as scalar loads like the Ft255 stream above
        using FT = LnField<NL == 4 ? FT127 : FT191>;
        constexpr int N = FT::N;
        const ConstU32* cidx = (const ConstU32*)a.colidx;
        const ConstU32* cvl = (const ConstU32*)a.vals29;
        for (u32 kb = k0; kb < k1; kb += 60) {
          const u32 ke = kb + 60 < k1 ? kb + 60 : k1;
          ln::LazyN<FT> acc;
          ln::lazy_zero<FT>(acc);
          u32 since = 0;
          Fe<NL> x = fe_load<NL>(xin + (size_t)cidx[kb] * pstride);
          u32 cn = cidx[kb + 1 < ke ? kb + 1 : kb];
          LN<N> v;
#pragma unroll
          for (int i = 0; i < N; i++) v.v[i] = cvl[(size_t)kb * FT::STRIDE + i];
          for (u32 k = kb; k < ke; k++) {
            Fe<NL> xn = x;
            mem_phase(true);
            if (k + 1 < ke) xn = fe_load<NL>(xin + (size_t)cn * pstride);
            const u32 kn = k + 1 < ke ? k + 1 : k, kn2 = k + 2 < ke ? k + 2 : k;
            cn = cidx[kn2];
            LN<N> vn;
#pragma unroll
            for (int i = 0; i < N; i++) vn.v[i] = cvl[(size_t)kn * FT::STRIDE + i];
            mem_phase(false);
            ln::lazy_mac<FT>(acc, ln::from_packed<FT>(x), v);
            v = vn;
            if (++since == 6) { ln::lazy_normalize<FT>(acc); since = 0; }
            x = xn;
          }
          res = fe_add<NL>(res, ln::lazy_reduce<FT>(acc));
        }
      }
    }
    if (!done) {
      for (u32 kb = k0; kb < k1; kb += 8) {
        Wide<NL> w = wide_zero<NL>();
        const u32 ke = kb + 8 < k1 ? kb + 8 : k1;
        for (u32 k = kb; k < ke; k++) {
          const u32 col = __builtin_amdgcn_readfirstlane(a.colidx[k]);
          const Fe<NL> v = fe_load<NL>(a.vals + (size_t)k * NL);
          wide_mac<NL>(w, v, fe_load<NL>(xin + (size_t)col * pstride));
        }
        res = fe_add<NL>(res, wide_reduce<NL>(w));
      }
    }
  }
  return res;
}

// OPW = outputs per workgroup (4: the wide levels; narrower ones go to spmm_t_sliced_kernel)
// n_main: the rows this launch covers, [0, n_main) (all of them, or the whole 64-row groups when spmm_t_tail_kernel takes the rest)
template <int NL, int SPMM_OPW>
__global__ void __launch_bounds__(128) spmm_t_kernel(SpmmTArgs a, u32 n_main) {
  const u64 row = (u64)blockIdx.y * 128 + threadIdx.x;
  if ((row & ~(u64)63) >= n_main) return;            // a wave with no row at all (<= 64 rows: the workgroup's second wave)
  const bool live = row < n_main;
  const u64 rr = live ? row : 0;                     // dead lanes recompute row 0 and do not store
  const u32* xin = a.t + (a.in_off * a.n_rows + rr) * NL;
  const size_t pstride = (size_t)a.n_rows * NL;      // words between consecutive positions
  for (u32 oo = 0; oo < SPMM_OPW; oo++) {
    const u64 o = (u64)blockIdx.x * SPMM_OPW + oo;
    if (o >= a.m) break;
    const u32 k0 = __builtin_amdgcn_readfirstlane(a.rowptr[o]);
    const u32 k1 = __builtin_amdgcn_readfirstlane(a.rowptr[o + 1]);
    const Fe<NL> res = spmm_t_terms<NL>(a, xin, pstride, k0, k1);
    if (live) {
      u32* dst = a.out_alt ? a.out_alt + (o * a.n_rows + row) * NL : a.t + ((a.out_off + o) * a.n_rows + row) * NL;
      fe_store<NL>(dst, res);
    }
  }
}

// Rows [n_main, n_rows) of every output, n_rows - n_main < 64: under the lane = row mapping they fill a fraction of a wave
// (C3's 101 rows: 37 of the second wave's 64 lanes, 21 % of all issued lanes idle in a kernel that VALU issue binds).  Here the
// lanes run over (output, tail row) pairs back to back, so that a wave holds the tails of two or three outputs and is full.
// The price: matrix entries are per lane (vector loads; 64 lanes share two or three distinct entries, L1 hits) instead of
// scalar, and a wave runs as long as its longest output.  Same dot products, same reduction points (every <= 60 terms).
__global__ void __launch_bounds__(256) spmm_t_tail_kernel(SpmmTArgs a, u32 n_main) {
  constexpr int NL = 8;
  const u32 tail = (u32)a.n_rows - n_main;
  const u64 total = a.m * tail;
  const u64 flat = (u64)blockIdx.x * 256 + threadIdx.x;
  if ((flat & ~(u64)63) >= total) return;
  const bool live = flat < total;
  const u64 f = live ? flat : total - 1;             // dead lanes shadow the last pair and do not store
  const u64 o = f / tail;
  const u32 row = n_main + (u32)(f - o * tail);
  const u32 k0 = a.rowptr[o], len = a.rowptr[o + 1] - k0;
  u32 maxlen = len;                                   // the wave's trip count
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) { const u32 other = (u32)__shfl_xor((int)maxlen, d, 64); maxlen = other > maxlen ? other : maxlen; }
  maxlen = __builtin_amdgcn_readfirstlane(maxlen);
  const u32* xin = a.t + (a.in_off * a.n_rows + row) * NL;
  const size_t pstride = (size_t)a.n_rows * NL;
  auto load_v = [&](u32 k) {
    Fe29 v;
    const uint4* p = reinterpret_cast<const uint4*>(a.vals29 + (size_t)k * 12);
    const uint4 lo = p[0], hi = p[1];
    v.v[0] = lo.x; v.v[1] = lo.y; v.v[2] = lo.z; v.v[3] = lo.w; v.v[4] = hi.x; v.v[5] = hi.y; v.v[6] = hi.z; v.v[7] = hi.w;
    v.v[8] = a.vals29[(size_t)k * 12 + 8];
    return v;
  };
  Fe<NL> res = fe_zero<NL>();
  Fe<NL> x = fe_zero<NL>();
  Fe29 v;
#pragma unroll
  for (int i = 0; i < 9; i++) v.v[i] = 0;
  if (len) { x = fe_load<NL>(xin + (size_t)a.colidx[k0] * pstride); v = load_v(k0); }
  for (u32 ib = 0; ib < maxlen; ib += 60) {
    const u32 ie = ib + 60 < maxlen ? ib + 60 : maxlen;
    Lazy29 acc;
    lazy29_zero(acc);
    u32 since = 0;
    for (u32 i = ib; i < ie; i++) {
      Fe<NL> xn = x;
      Fe29 vn = v;
      mem_phase(true);
      if (i + 1 < len) { xn = fe_load<NL>(xin + (size_t)a.colidx[k0 + i + 1] * pstride); vn = load_v(k0 + i + 1); }
      mem_phase(false);
      if (i < len) lazy29_mac(acc, fe_to29(x), v);
      if (++since == 6) { lazy29_normalize(acc); since = 0; }
      x = xn; v = vn;
    }
    res = fe_add<NL>(res, lazy29_reduce(acc));
  }
  if (live) {
    u32* dst = a.out_alt ? a.out_alt + (o * a.n_rows + row) * NL : a.t + ((a.out_off + o) * a.n_rows + row) * NL;
    fe_store<NL>(dst, res);
  }
}

// The tail levels of the recursion have a few hundred outputs or fewer, each a dependent chain of up to ~100 gathered
// terms: the launch is bound by that chain's latency, not by throughput.  SL wave-groups per output each take 1/SL of the
// terms; the partial sums (field elements: addition is exact, so the split does not change the result) meet in LDS.
template <int NL, int SL>
__global__ void __launch_bounds__(128 * SL) spmm_t_sliced_kernel(SpmmTArgs a) {
  __shared__ u32 part[(SL - 1) * 128 * NL];
  const u32 lane = threadIdx.x & 127;
  const u32 sl = __builtin_amdgcn_readfirstlane(threadIdx.x >> 7);
  const u64 row = (u64)blockIdx.y * 128 + lane;
  const bool live = row < a.n_rows;
  const u64 rr = live ? row : 0;
  const u32* xin = a.t + (a.in_off * a.n_rows + rr) * NL;
  const size_t pstride = (size_t)a.n_rows * NL;
  const u64 o = blockIdx.x;
  const u32 k0 = __builtin_amdgcn_readfirstlane(a.rowptr[o]);
  const u32 k1 = __builtin_amdgcn_readfirstlane(a.rowptr[o + 1]);
  const u32 len = k1 - k0;
  const u32 ks = k0 + (u32)(((u64)len * sl) / SL), ke = k0 + (u32)(((u64)len * (sl + 1)) / SL);
  const bool wave_live = (row & ~(u64)63) < a.n_rows;  // (a wave with no row at all still has to reach the barrier)
  Fe<NL> res = wave_live ? spmm_t_terms<NL>(a, xin, pstride, ks, ke) : fe_zero<NL>();
  if (sl) {
#pragma unroll
    for (int i = 0; i < NL; i++) part[((sl - 1) * NL + i) * 128 + lane] = res.v[i];
  }
  __syncthreads();
  if (sl == 0) {
#pragma unroll
    for (int s = 0; s < SL - 1; s++) {
      Fe<NL> p;
#pragma unroll
      for (int i = 0; i < NL; i++) p.v[i] = part[(s * NL + i) * 128 + lane];
      res = fe_add<NL>(res, p);
    }
    if (live) {
      u32* dst = a.out_alt ? a.out_alt + (o * a.n_rows + row) * NL : a.t + ((a.out_off + o) * a.n_rows + row) * NL;
      fe_store<NL>(dst, res);
    }
  }
}
hipError_t launch_spmm_t(int nl, const SpmmTArgs& a, hipStream_t st) {
  if (a.m == 0 || a.n_rows == 0) return hipSuccess;
  if (a.m >= 8192) {
    // a last group of <= 48 rows goes to the packed-tail kernel (Ft255 limb path), the whole 64-row groups stay lane = row
    u32 n_main = (u32)a.n_rows;
    const u32 tail = (u32)(a.n_rows & 63);
    if (nl == 8 && a.vals29 != nullptr && tail != 0 && tail <= 48) n_main -= tail;
    if (n_main) {
      dim3 grid((unsigned)((a.m + 3) / 4), (unsigned)((n_main + 127) / 128));
      LCPC_DISPATCH_NL(nl, hipLaunchKernelGGL((spmm_t_kernel<NLV, 4>), grid, dim3(128), 0, st, a, n_main));
    }
    if (n_main != (u32)a.n_rows) {
      const u64 total = a.m * (a.n_rows - n_main);
      const dim3 tg((unsigned)((total + 255) / 256));
      hipLaunchKernelGGL(spmm_t_tail_kernel, tg, dim3(256), 0, st, a, n_main);
    }
  } else if (a.m > 2048) {
    dim3 grid((unsigned)a.m, (unsigned)((a.n_rows + 127) / 128));
    LCPC_DISPATCH_NL(nl, hipLaunchKernelGGL((spmm_t_sliced_kernel<NLV, 2>), grid, dim3(256), 0, st, a));
  } else if (a.m > 256) {
    dim3 grid((unsigned)a.m, (unsigned)((a.n_rows + 127) / 128));
    LCPC_DISPATCH_NL(nl, hipLaunchKernelGGL((spmm_t_sliced_kernel<NLV, 4>), grid, dim3(512), 0, st, a));
  } else {
    dim3 grid((unsigned)a.m, (unsigned)((a.n_rows + 127) / 128));
    LCPC_DISPATCH_NL(nl, hipLaunchKernelGGL((spmm_t_sliced_kernel<NLV, 8>), grid, dim3(1024), 0, st, a));
  }
  return hipGetLastError();
}

template <int NL>
__global__ void __launch_bounds__(128) sdig_rs_t_kernel(const u32* in_t, u32 n_in, u32* t, u64 out_off, u32 n_out, u64 n_rows,
                                                       const u32* r2) {
  const u64 row = (u64)blockIdx.y * 128 + threadIdx.x;
  if (row >= n_rows) return;
  const u32 k = blockIdx.x;
  Fe<NL> raw = fe_zero<NL>();
  raw.v[0] = k + 1;
  const Fe<NL> x = fe_mul<NL>(raw, fe_load<NL>(r2));          // (k+1) in Montgomery form
  Fe<NL> r = fe_zero<NL>();
  for (u32 j = n_in; j-- > 0;) r = fe_add<NL>(fe_mul<NL>(r, x), fe_load<NL>(in_t + ((u64)j * n_rows + row) * NL));
  fe_store<NL>(t + ((out_off + k) * n_rows + row) * NL, r);
}
hipError_t launch_sdig_rs_t(int nl, const u32* in_t, u32 n_in, u32* t, u64 out_off, u32 n_out, u64 n_rows, const u32* r2,
                            hipStream_t st) {
  if (!n_out || !n_rows) return hipSuccess;
  dim3 grid(n_out, (unsigned)((n_rows + 127) / 128));
  LCPC_DISPATCH_NL(nl, hipLaunchKernelGGL(sdig_rs_t_kernel<NLV>, grid, dim3(128), 0, st, in_t, n_in, t, out_off, n_out, n_rows, r2));
  return hipGetLastError();
}

// =================================================================================================
// precomp_fft on the device: roots[i] = w^i, i < n/2, from the host-supplied squares
// pw[j] = w^(2^j) (Montgomery).  Thread i multiplies the squares selected by the bits of i (<= log n
// products); for Ft255 it also emits the 29-bit-limb / 2^261 form used by fe_mul_r29.
// =================================================================================================
template <int NL>
__global__ void __launch_bounds__(256) roots_kernel(const u32* pw, u32 log_half, const u32* one, u32* roots, u32* roots29,
                                                   u32* roots29c) {
  const u64 i = (u64)blockIdx.x * 256 + threadIdx.x;
  if (i >= ((u64)1 << log_half)) return;
  Fe<NL> acc = fe_load<NL>(one);
  for (u32 j = 0; j < log_half; j++)
    if ((i >> j) & 1) acc = fe_mul<NL>(acc, fe_load<NL>(pw + (size_t)j * NL));
  fe_store<NL>(roots + i * NL, acc);
  if constexpr (NL == 8) {
    if (roots29 != nullptr) {
      Fe<8> t = acc;
#pragma unroll
      for (int d = 0; d < 5; d++) t = fe_add<8>(t, t);          // * 2^5: R = 2^256 -> 2^261
      const Fe29 x = fe_to29(t);
#pragma unroll
      for (int k = 0; k < 9; k++) roots29[i * 12 + k] = x.v[k];
      roots29[i * 12 + 9] = roots29[i * 12 + 10] = roots29[i * 12 + 11] = 0;
    }
    if (roots29c != nullptr) {                                   // w^i * 2^5: the converting table of ntt_pass_l9_kernel
      Fe<8> t = fe_canon<8>(acc);
#pragma unroll
      for (int d = 0; d < 5; d++) t = fe_add<8>(t, t);
      const Fe29 x = fe_to29(t);
#pragma unroll
      for (int k = 0; k < 9; k++) roots29c[i * 12 + k] = x.v[k];
      roots29c[i * 12 + 9] = roots29c[i * 12 + 10] = roots29c[i * 12 + 11] = 0;
    }
  }
}
hipError_t launch_roots(int nl, const u32* pw, u32 log_half, const u32* one, u32* roots, u32* roots29, u32* roots29c,
                        hipStream_t st) {
  const u64 n = (u64)1 << log_half;
  dim3 grid((unsigned)((n + 255) / 256));
  LCPC_DISPATCH_NL(nl, hipLaunchKernelGGL(roots_kernel<NLV>, grid, dim3(256), 0, st, pw, log_half, one, roots, roots29, roots29c));
  return hipGetLastError();
}

template <int NL>
__global__ void __launch_bounds__(256) pad_rows_kernel(const u32* src, u64 src_stride, u32* dst, u64 dst_stride, u64 n_valid) {
  const u64 row = blockIdx.y;
  for (u64 e = (u64)blockIdx.x * 256 + threadIdx.x; e < n_valid; e += (u64)gridDim.x * 256)
    fe_store<NL>(dst + (row * dst_stride + e) * NL, fe_load<NL>(src + (row * src_stride + e) * NL));
}
hipError_t launch_pad_rows(int nl, const u32* src, u64 src_stride, u32* dst, u64 dst_stride, u64 n_valid, u64 n_rows,
                           hipStream_t st) {
  if (n_rows == 0 || n_valid == 0) return hipSuccess;
  if (n_rows > 65535) return hipErrorInvalidValue;
  unsigned gx = (unsigned)((n_valid + 255) / 256);
  if (gx > 1024) gx = 1024;
  dim3 grid(gx, (unsigned)n_rows);
  switch (nl) {
    case 2: hipLaunchKernelGGL(pad_rows_kernel<2>, grid, dim3(256), 0, st, src, src_stride, dst, dst_stride, n_valid); break;
    case 4: hipLaunchKernelGGL(pad_rows_kernel<4>, grid, dim3(256), 0, st, src, src_stride, dst, dst_stride, n_valid); break;
    case 6: hipLaunchKernelGGL(pad_rows_kernel<6>, grid, dim3(256), 0, st, src, src_stride, dst, dst_stride, n_valid); break;
    case 8: hipLaunchKernelGGL(pad_rows_kernel<8>, grid, dim3(256), 0, st, src, src_stride, dst, dst_stride, n_valid); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

}  // namespace lcpc
