// lcpc_amd/csrc/ntt_u.hip -- K1u: the Ft255 row NTT with LANE = ROW and wave-uniform twiddles (LcEncoding::encode for Ligero,
// lcpc-ligero-pc/src/lib.rs:162-164 = fffft fft_io_pc [3P]), for commitments of >= 64 rows at 2^18 columns (the headline shape).
//
// Every twiddle of the row NTT is shared by all rows of the matrix.  K1s (ntt_l9s.hip) gives a lane one quad of ONE row, so its
// twiddles differ per lane and each multiply is a full Montgomery product: 153 mads + 35.  Here a wave's 64 lanes are 64 ROWS at
// the same positions: the twiddle w is wave-uniform, and x * w mod p becomes  sum_j x_j * W_j  with the nine precomputed
// constants W_j = balanced(w 2^(29 j) mod p) read as SCALAR operands, plus one 32-bit quotient: 90 mads + 27
// (gen/gen_wmul_asm.py).  Measured in isolation (tools/ubench_wmul.hip, profiles/r05_ubench_wmul.jsonl): 1.6-1.8 x the
// multiplies per second of the Montgomery form.
//
// What lane = row costs: a workgroup holds 64 rows x 64 positions (4096 elements, as K1s's 4 x 1024), so a pass covers 6 stages
// instead of 8-10 and the 18 stages take THREE passes over HBM (6 + 6 + 6) with a position-major intermediate
// mid[row group][position][64 rows] (every access of it a 2 KiB run, whatever the stride between positions):
//   pass 0  stages 0..5   positions i * 4096 + lo   row-major source (+ the LcCommit.coeffs copy) -> mid
//   pass 1  stages 6..11  positions hi * 4096 + i * 64 + lo                                   mid -> mid (in place)
//   pass 2  stages 12..17 positions hi * 64 + i      mid -> row-major canonical comm (128-byte runs per row)
// The tile lives in REGISTERS -- a thread keeps the eight elements of its two radix-4 quads (72 VGPRs) through the pass -- and
// LDS only carries the regrouping between the three radix-4 rounds, in three slices (limbs 0-3 | 4-7 | 8: 64 KiB at a time), so
// that two workgroups of 512 threads fit a CU and one's loads and stores run under the other's arithmetic.
//
// Same butterflies, same canonical-output trick ("block 0" multiplies take the converting constants) and the same lazy signed
// 9 x 29-bit limbs as K1s; exact modular arithmetic, so the fully reduced bits equal the reference's radix-2 loop.
#include <algorithm>
#include "kernels.h"
#include "ntt_l9_dev.h"

namespace lcpc {
#include "field_wmul_gen.h"   // wmul_u(): the one-statement shifted-multiples multiply

namespace {

constexpr u32 U_SLOT = 96;                                  // words per twiddle slot (81 used; 384-byte stride)
constexpr u32 U_QUAD = 4 * U_SLOT;                          // slots of a quad: w0 | w1 | w2 (for c3) | w2 for c1
constexpr u32 U_CLASS = 3 * 16 * U_QUAD;                    // rounds x quads: 18432 words = 72 KiB per tile class

struct NttUArgs {
  const u32* src;         // pass 0: row-major (src_stride elements per row); passes 1, 2: mid
  u32* dst;               // passes 0, 1: mid; pass 2: row-major (dst_stride elements per row)
  const u32* pack;        // this pass's twiddle pack: [class][round][quad][slot][U_SLOT]
  const u32* qp29;        // (i - 24) p table of l9::clamp
  u32* copy_dst;          // pass 0, may be null: padded copy of src (LcCommit.coeffs)
  u64 src_stride, dst_stride, n_valid, n_src_total, n_rows;
  u32 log_n, n_groups;
};

// element index inside the 64-position tile of slot c of quad q in round r (stages 6 pass + 2 r, + 1): distance 16, 4, 1
template <int R> LCPC_DEV u32 u_idx(u32 q, u32 c) {
  if constexpr (R == 0) return q + 16u * c;
  else if constexpr (R == 1) return ((q >> 2) << 4) + 4u * c + (q & 3u);
  else return 4u * q + c;
}

LCPC_DEV L9 u_mul(const L9& x, const u32* np2, const u32* w) {
  L9 r;
  wmul_u(x.v, np2, w, r.v);
  return r;
}

// regroup the eight elements of a thread from round R's quads to round R + 1's through LDS, 4 + 4 + 1 limbs at a time
template <int R> LCPC_DEV void u_exchange(L9 (&E)[2][4], u32* xch, const u32 (&qa)[2], u32 lane) {
  uint4* x4 = reinterpret_cast<uint4*>(xch);
#pragma unroll
  for (int half = 0; half < 2; half++) {
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
      for (int c = 0; c < 4; c++)
        x4[u_idx<R>(qa[a], c) * 64 + lane] = make_uint4(E[a][c].v[4 * half], E[a][c].v[4 * half + 1], E[a][c].v[4 * half + 2], E[a][c].v[4 * half + 3]);
    __syncthreads();
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
      for (int c = 0; c < 4; c++) {
        const uint4 t = x4[u_idx<R + 1>(qa[a], c) * 64 + lane];
        E[a][c].v[4 * half] = t.x; E[a][c].v[4 * half + 1] = t.y; E[a][c].v[4 * half + 2] = t.z; E[a][c].v[4 * half + 3] = t.w;
      }
    __syncthreads();
  }
#pragma unroll
  for (int a = 0; a < 2; a++)
#pragma unroll
    for (int c = 0; c < 4; c++) xch[u_idx<R>(qa[a], c) * 64 + lane] = E[a][c].v[8];
  __syncthreads();
#pragma unroll
  for (int a = 0; a < 2; a++)
#pragma unroll
    for (int c = 0; c < 4; c++) E[a][c].v[8] = xch[u_idx<R + 1>(qa[a], c) * 64 + lane];
  __syncthreads();
}

// one radix-4 butterfly (stages t, t + 1) on x0..x3 at distance D, as K1s's: outputs in place.  w: the quad's four slots.
// Value ranges (u_mul returns (-2p, 2.7p)): inputs |x| < 5.4p -> b < 10.8p, c0 < 21.6p (inside the clamp table's -24p .. 39p);
// the difference b0 - b1 has limbs up to 2^30 and is normalised before its multiply (the quotient estimate wants sum |limb| <
// 9 * 2^29); x0 - x2, x1 - x3 and b2 - b3 are differences of normalised values.
template <bool LAST_TWO> LCPC_DEV void u_butterfly(L9 (&x)[4], const u32* np2, const u32* nqp, const u32* w) {
  if constexpr (LAST_TWO) {
    // stages k-2, k-1: twiddles 1, w^(n/4), 1 -- slot 1 holds w^(n/4)
    const L9 b0 = l9::add(x[0], x[2]), b1 = l9::add(x[1], x[3]);
    const L9 b2 = l9::sub(x[0], x[2]);
    const L9 b3 = u_mul(l9::sub(x[1], x[3]), np2, w + U_SLOT);
    x[0] = l9::add(b0, b1); x[1] = l9::sub(b0, b1); x[2] = l9::add(b2, b3); x[3] = l9::sub(b2, b3);
    l9::normalize(x[0]); l9::normalize(x[1]); l9::normalize(x[2]); l9::normalize(x[3]);
  } else {
    const L9 b0 = l9::add(x[0], x[2]), b1 = l9::add(x[1], x[3]);
    L9 c0 = l9::add(b0, b1);
    l9::clamp_apply(c0, l9::clamp_row(nqp, l9::clamp_q(c0.v[8])));                        // [0, p + 2^239)
    L9 d1 = l9::sub(b0, b1);
    l9::normalize(d1);
    const L9 b2 = u_mul(l9::sub(x[0], x[2]), np2, w);
    const L9 b3 = u_mul(l9::sub(x[1], x[3]), np2, w + U_SLOT);
    const L9 c1 = u_mul(d1, np2, w + 3 * U_SLOT);
    L9 c2 = l9::add(b2, b3);
    l9::normalize(c2);
    const L9 c3 = u_mul(l9::sub(b2, b3), np2, w + 2 * U_SLOT);
    x[0] = c0; x[1] = c1; x[2] = c2; x[3] = c3;
  }
}

template <int PASS>
__global__ void __launch_bounds__(512, 4) ntt_u_kernel(NttUArgs a) {
  extern __shared__ __attribute__((aligned(16))) u32 lds[];
  u32* xch = lds;                                           // 64 positions x 64 lanes x 16 bytes
  u32* nqp = lds + 64 * 64 * 4;                             // negated (i - 24) p rows, 12-word stride
  const u32 k = a.log_n;
  const u32 G = a.n_groups;
  // XCD-aware order (workgroup b runs on XCD b mod 8): the row groups of one tile back to back on one XCD, so that the tile
  // class's 72 KiB of twiddle constants are fetched from HBM once and then come out of that XCD's L2
  const u32 xcd = blockIdx.x & 7u;
  const u32 qq = blockIdx.x >> 3;
  u32 tile, g;
  if constexpr (PASS == 0) {
    // the first pass gathers single 32-byte elements of a row (positions lo + 4096 i): the four tiles lo .. lo + 3 that share
    // each 128-byte line run as consecutive workgroups of one XCD, so that the line is fetched from HBM once and the three
    // other tiles find it in that L2 (and their partial-line stores of the coeffs copy meet there)
    const u32 sub = qq & 3u, q2 = qq >> 2;
    tile = (((q2 / G) * 8u + xcd) << 2) | sub;
    g = q2 % G;
  } else {
    tile = (qq / G) * 8u + xcd;
    g = qq % G;
  }
  const u32 stride_log = k - 6u * PASS - 6u;                // log2 of the distance between the tile's positions
  u32 hi, lo, cls;
  if constexpr (PASS == 0) { hi = 0; lo = tile; cls = tile; }
  else if constexpr (PASS == 1) { lo = tile >> 6; hi = tile & 63u; cls = lo + (hi == 0 ? 64u : 0u); }
  else { hi = tile; lo = 0; cls = hi == 0 ? 1u : 0u; }
  const u32 base = (hi << (stride_log + 6u)) | lo;
  const u32 tid = threadIdx.x, lane = tid & 63u;
  const u32 wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const u32 qa[2] = {wv, wv + 8u};
  const u64 row = (u64)g * 64 + lane;
  const bool live = row < a.n_rows;
  for (u32 i = tid; i < 64 * 12; i += 512) nqp[i] = 0u - a.qp29[i];

  u32 np2[9];
#pragma unroll
  for (int j = 0; j < 9; j++) np2[j] = 0u - 2u * (u32)P29::limb(j);

  // ---- load: the thread's two quads of round 0 ----
  L9 E[2][4];
#pragma unroll
  for (int q2 = 0; q2 < 2; q2++)
#pragma unroll
    for (int c = 0; c < 4; c++) {
      const u32 pos = base + (u_idx<0>(qa[q2], c) << stride_log);
      Fe<8> v = fe_zero<8>();
      if constexpr (PASS == 0) {
        const u64 flat = row * a.src_stride + pos;
        if (live && pos < a.n_valid && flat < a.n_src_total) {
          v = fe_load<8>(a.src + flat * 8);
          if (a.copy_dst != nullptr) fe_store<8>(a.copy_dst + flat * 8, v);
        } else if (live && a.copy_dst != nullptr && pos < a.n_valid) {
          fe_store<8>(a.copy_dst + flat * 8, v);            // the zero tail of a ragged last row
        }
      } else {
        v = fe_load<8>(a.src + ((((u64)g << k) + pos) * 64 + lane) * 8);
      }
      E[q2][c] = l9::from_packed(v);
    }
  __syncthreads();                                           // nqp is in place

  const u32* cpk = a.pack + (size_t)cls * U_CLASS;
  // ---- three radix-4 rounds ----
#pragma unroll
  for (int q2 = 0; q2 < 2; q2++) u_butterfly<false>(E[q2], np2, nqp, cpk + (0 * 16 + qa[q2]) * U_QUAD);
  u_exchange<0>(E, xch, qa, lane);
#pragma unroll
  for (int q2 = 0; q2 < 2; q2++) u_butterfly<false>(E[q2], np2, nqp, cpk + (1 * 16 + qa[q2]) * U_QUAD);
  u_exchange<1>(E, xch, qa, lane);
#pragma unroll
  for (int q2 = 0; q2 < 2; q2++) u_butterfly<PASS == 2>(E[q2], np2, nqp, cpk + (2 * 16 + qa[q2]) * U_QUAD);

  // ---- store: round 2's quads hold four consecutive tile positions each ----
#pragma unroll
  for (int q2 = 0; q2 < 2; q2++)
#pragma unroll
    for (int c = 0; c < 4; c++) {
      const u32 i = u_idx<2>(qa[q2], c);
      const u32 pos = base + (i << stride_log);
      L9 x = E[q2][c];                                                  // normalised, |value| < 22p
      l9::clamp_apply(x, l9::clamp_row(nqp, l9::clamp_q(x.v[8])));      // [0, p + 2^239) < 2^256
      u32 w8[8];
      fe_from29(w8, x.v);
      Fe<8> v;
#pragma unroll
      for (int t = 0; t < 8; t++) v.v[t] = w8[t];
      if constexpr (PASS == 2) {
        if (__any((int)(x.v[8] >= (u32)P29::limb(8)))) v = fe_reduce_once8(w8);   // -> [0, p): about one element in 2^17 needs it
        if (hi == 0 && i < 4) v = fe_canon_r29(v);                        // the never-multiplied prefix of a row is still in Montgomery form
        if (live) fe_store<8>(a.dst + (row * a.dst_stride + pos) * 8, v);
      } else {
        fe_store<8>(a.dst + ((((u64)g << k) + pos) * 64 + lane) * 8, v);
      }
    }
}

// ---- the twiddle packs --------------------------------------------------------------------------------------------------------
// One thread per (class, round, quad, slot): the table index of that twiddle, plain or converting, and its nine shifted multiples
// W_j = balanced(w 2^(29 j) mod p) as 81 words t = 9 k + j (limb k of W_j).  roots[i] = w^i R (Montgomery, R = 2^256):
// mont_mul(w^i R, 2^(29 j)) = w^i 2^(29 j) is the plain constant; the converting one (a "block 0" multiply turns Montgomery-form
// data into canonical values on the fly, ntt_pass_l9_kernel's trick) is mont_mul(mont_mul(w^i R, 1), 2^(29 j)) = w^i 2^(29 j) R^-1.
__global__ void __launch_bounds__(256) ntt_u_pack_kernel(const u32* roots, u32 log_n, u32 pass, u32 n_classes, u32* pack) {
  const u32 k = log_n;
  const u64 total = (u64)n_classes * 3 * 16 * 4;
  const u32 stride_log = k - 6u * pass - 6u;
  for (u64 id = (u64)blockIdx.x * 256 + threadIdx.x; id < total; id += (u64)gridDim.x * 256) {
    const u32 slot = (u32)(id & 3), q = (u32)((id >> 2) & 15), r = (u32)((id >> 6) % 3), cls = (u32)(id / 192);
    u32 lo, hi0;                                             // hi0: the class of the tiles with hi == 0 (they hold block 0)
    if (pass == 0) { lo = cls; hi0 = 1; }
    else if (pass == 1) { lo = cls & 63u; hi0 = cls >= 64; }
    else { lo = 0; hi0 = cls == 1; }
    const u32 t = 6u * pass + 2u * r;
    const u32 i0 = r == 0 ? q : (r == 1 ? ((q >> 2) << 4) + (q & 3u) : 4u * q);
    const u32 di = 16u >> (2u * r);                          // distance of the quad's slots in tile positions
    const u32 g0 = (i0 << stride_log) | lo, g1 = g0 + (di << stride_log);
    const bool last_two = t + 2 == k;
    u32 widx;
    bool conv = false;
    if (last_two) {
      widx = 1u << (k - 2);                                  // w^(n/4), plain (its inputs are canonical, or converted at the store)
    } else {
      const u32 gm0 = (1u << (k - t - 1)) - 1, gm1 = gm0 >> 1;
      widx = slot == 0 ? (g0 & gm0) << t : (slot == 1 ? (g1 & gm0) << t : (g0 & gm1) << (t + 1));
      const bool blk0 = hi0 && i0 < (16u >> (2u * r));       // all four inputs never multiplied so far
      conv = blk0 && slot != 2;                              // w0, w1 and the w2 of c1 leave block 0; c3's inputs are canonical already
    }
    Fe<8> w = fe_load<8>(roots + (size_t)widx * 8);
    if (conv) { Fe<8> one = fe_zero<8>(); one.v[0] = 1; w = fe_mul<8>(w, one); }
    Fe<8> sh = fe_zero<8>();                                 // 2^(29 j) as a plain integer
    u32* out = pack + ((((size_t)cls * 3 + r) * 16 + q) * 4 + slot) * U_SLOT;
    for (u32 j = 0; j < 9; j++) {
#pragma unroll
      for (int z = 0; z < 8; z++) sh.v[z] = 0;
      sh.v[(29 * j) / 32] = 1u << ((29 * j) % 32);
      const Fe<8> v = fe_mul<8>(w, sh);                      // in [0, p)
      // balanced: v > (p - 1) / 2 -> v - p, as a 288-bit two's complement number
      u32 m[9];
#pragma unroll
      for (int z = 0; z < 8; z++) m[z] = v.v[z];
      m[8] = 0;
      bool big = false, decided = false;
#pragma unroll
      for (int z = 7; z >= 0; z--) {
        const u32 hz = (Mod<8>::P[z] >> 1) | (z < 7 ? (Mod<8>::P[z + 1] & 1u) << 31 : 0u);     // limb z of (p - 1) / 2
        if (!decided && v.v[z] != hz) { big = v.v[z] > hz; decided = true; }
      }
      if (big) {
        u64 br = 0;
#pragma unroll
        for (int z = 0; z < 9; z++) {
          const u64 d = (u64)m[z] - (z < 8 ? Mod<8>::P[z] : 0u) - br;
          m[z] = (u32)d;
          br = (d >> 32) & 1u;
        }
      }
      for (u32 kk = 0; kk < 9; kk++) {
        const u32 b = 29 * kk, wd = b / 32, shb = b % 32;
        u64 x = (u64)m[wd] >> shb;
        if (wd + 1 < 9) x |= (u64)m[wd + 1] << (32 - shb);
        const u32 limb = kk < 8 ? (u32)(x & P29::M) : (u32)x;   // limb 8: bits 232 .. 263, i.e. sign-extended (m[8] is 0 or ~0)
        out[9 * kk + j] = limb;
      }
    }
    for (u32 z = 81; z < U_SLOT; z++) out[z] = 0;
  }
}

}  // namespace

bool ntt_u_supported(uint32_t log_n) { return log_n == 18; }
uint32_t ntt_u_classes(uint32_t log_n, uint32_t pass) { return pass == 0 ? 1u << (log_n - 6) : (pass == 1 ? 128u : 2u); }
uint64_t ntt_u_pack_words(uint32_t log_n, uint32_t pass) { return (uint64_t)ntt_u_classes(log_n, pass) * U_CLASS; }

hipError_t launch_ntt_u_pack(const uint32_t* roots, uint32_t log_n, uint32_t pass, uint32_t* pack, hipStream_t st) {
  const u32 n_classes = ntt_u_classes(log_n, pass);
  const u64 total = (u64)n_classes * 192;
  const unsigned blocks = (unsigned)std::min<u64>((total + 255) / 256, 65536);
  hipLaunchKernelGGL(ntt_u_pack_kernel, dim3(blocks), dim3(256), 0, st, roots, log_n, pass, n_classes, pack);
  return hipGetLastError();
}

// a: src / dst row-major (src_stride / dst_stride), n_valid, n_src_total, copy_dst, n_rows, log_n, qp29 as for the other Ft255 kernels;
// mid: n_groups x n_cols x 64 x 32 bytes; packs: the three passes' packs
hipError_t launch_ntt_u(const NttPassArgs& p, uint32_t* mid, const uint32_t* const packs[3], hipStream_t st) {
  NttUArgs a{};
  a.qp29 = p.qp29; a.src_stride = p.src_stride; a.dst_stride = p.dst_stride; a.n_valid = p.n_valid; a.n_src_total = p.n_src_total;
  a.n_rows = p.n_rows; a.log_n = p.log_n; a.n_groups = (u32)((p.n_rows + 63) / 64);
  const unsigned grid = (unsigned)(((u64)1 << (p.log_n - 6)) * a.n_groups);
  const size_t lds = (size_t)(64 * 64 * 4 + 64 * 12) * 4;
  hipError_t e;
  if ((e = hipFuncSetAttribute(reinterpret_cast<const void*>(&ntt_u_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)) != hipSuccess) return e;
  if ((e = hipFuncSetAttribute(reinterpret_cast<const void*>(&ntt_u_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)) != hipSuccess) return e;
  if ((e = hipFuncSetAttribute(reinterpret_cast<const void*>(&ntt_u_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)) != hipSuccess) return e;
  a.src = p.src; a.dst = mid; a.pack = packs[0]; a.copy_dst = p.copy_dst;
  hipLaunchKernelGGL(ntt_u_kernel<0>, dim3(grid), dim3(512), lds, st, a);
  a.src = mid; a.dst = mid; a.pack = packs[1]; a.copy_dst = nullptr;
  hipLaunchKernelGGL(ntt_u_kernel<1>, dim3(grid), dim3(512), lds, st, a);
  a.src = mid; a.dst = p.dst; a.pack = packs[2];
  hipLaunchKernelGGL(ntt_u_kernel<2>, dim3(grid), dim3(512), lds, st, a);
  return hipGetLastError();
}

}  // namespace lcpc
