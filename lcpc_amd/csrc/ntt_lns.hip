// lcpc_amd/csrc/ntt_lns.hip -- K1n: the shape-specialised lazy-limb row NTT for Ft63 / Ft127 / Ft191
// (LcEncoding::encode for Ligero, lcpc-ligero-pc/src/lib.rs:162-164 = fffft fft_io_pc [3P]; the reference benches Ft127
// besides Ft255: lcpc-ligero-pc/src/bench.rs:178-204), for the two-pass plans on 1024-element tiles (2^11 <= n_cols <= 2^20).
//
// The structure is that of K1s (ntt_l9s.hip, Ft255): radix-4 DIF rounds on a tile that lives in LDS in the multiplier's own
// format -- here N signed limbs of W bits (field_ln.h: 3 x 26, 5 x 29, 7 x 29) --, the pass shape as template parameters,
// twiddles from lane-order packs, normalise + clamp of the pure-sum output in one carry pass, an odd stage count peeled as
// a radix-2 round at stage 0 where a zero-padded row needs no additions.  What the smaller fields change:
//   * occupancy: 12 / 20 / 28 bytes of LDS per element and 64 / 72 / 96 VGPRs give 8 / 7 / 5 waves per SIMD;
//   * canonical output as in K1s (a.roots29c != null, the commit paths): butterflies in "block 0" -- the elements no
//     twiddle has touched yet -- take their twiddles from the converting set w^i R' R^-1, everything else is canonical
//     already, the 4 elements per row that only meet trivial twiddles are reduced at the store; the column hash then
//     reads comm as it is (its per-element Montgomery reduction was 16-28 % of the leaf kernel for these fields);
//   * the first pass stores values in [0, p + 64 B) < 2^(32 NL) without the final conditional subtract, the last pass
//     subtracts under a wave-level __any (Ft63 / Ft127: most waves; Ft191: rarely).
// The radix-4 butterfly is K1s's true radix-4 form (ntt_l9s.hip, head comment): w1 = I w0 with I = w^(n/4) the field's fixed 4th
// root of unity, so t = (x1 - x3) I, c2 = ((x0 - x2) + t) w0, c3 = ((x0 - x2) - t) w^(3e): three lane-varying multiplies and one by a
// constant every lane shares -- the shifted-multiples multiply with scalar operands for Ft127 / Ft191 (ln::mul_u), the ordinary
// one on a broadcast table entry for Ft63, whose 3-limb Montgomery multiply is the shorter of the two.
// Round 6 also brought over from K1s: the uniform rounds (Ft127 / Ft191: Shape::RU, the swizzled tile, block 0 converted one round
// earlier, so that for these two fields nothing is left in Montgomery form at the store -- only Ft63 still has the 4-element prefix),
// the wave priorities (field_dev.h mem_phase), the first round run from the thread's own loads, and -- Ft63 / Ft127 -- the final
// round of a last pass storing its four consecutive elements itself.
// Exact modular arithmetic: any stage grouping gives the same fully-reduced bits as the reference's radix-2 loop.
// The general kernel (kernels.hip ntt_pass_kernel) remains for one-pass rows, plans whose tables do not fit and as the A/B reference
// (LCPC_NTT_GENERAL=1; tests/test_gpu_ntt_shapes.py).
#include "kernels.h"
#include "field_ln.h"

namespace lcpc {

namespace {

// ---- an array of `cnt` elements in planes: limbs 0-3 as uint4 (N >= 5), then a uint2 plane (N = 3: limbs 0-1; N = 7:
//      limbs 4-5), then the top limb as u32.  Used for the LDS tile (cnt = 1024: unit-stride lanes are conflict-free in
//      every plane) and for the twiddle packs (cnt = variants x period) -------------------------------------------------
template <class FT> LCPC_DEV LN<FT::N> planes_get(const u32* base, u32 cnt, u32 e) {
  constexpr int N = FT::N;
  LN<N> r;
  if constexpr (N == 3) {
    const uint2 a = *reinterpret_cast<const uint2*>(base + (size_t)e * 2);
    r.v[0] = a.x; r.v[1] = a.y; r.v[2] = base[(size_t)cnt * 2 + e];
  } else {
    const uint4 a = *reinterpret_cast<const uint4*>(base + (size_t)e * 4);
    r.v[0] = a.x; r.v[1] = a.y; r.v[2] = a.z; r.v[3] = a.w;
    if constexpr (N == 5) r.v[4] = base[(size_t)cnt * 4 + e];
    else {
      const uint2 b = *reinterpret_cast<const uint2*>(base + (size_t)cnt * 4 + (size_t)e * 2);
      r.v[4] = b.x; r.v[5] = b.y; r.v[6] = base[(size_t)cnt * 6 + e];
    }
  }
  return r;
}
template <class FT> LCPC_DEV void planes_put(u32* base, u32 cnt, u32 e, const LN<FT::N>& x) {
  constexpr int N = FT::N;
  if constexpr (N == 3) {
    *reinterpret_cast<uint2*>(base + (size_t)e * 2) = make_uint2(x.v[0], x.v[1]);
    base[(size_t)cnt * 2 + e] = x.v[2];
  } else {
    *reinterpret_cast<uint4*>(base + (size_t)e * 4) = make_uint4(x.v[0], x.v[1], x.v[2], x.v[3]);
    if constexpr (N == 5) base[(size_t)cnt * 4 + e] = x.v[4];
    else {
      *reinterpret_cast<uint2*>(base + (size_t)cnt * 4 + (size_t)e * 2) = make_uint2(x.v[4], x.v[5]);
      base[(size_t)cnt * 6 + e] = x.v[6];
    }
  }
}
// one entry of the limb-form twiddle table / of the q*p table (STRIDE words per entry)
template <class FT> LCPC_DEV LN<FT::N> tab_entry(const u32* tab, u32 idx) {
  LN<FT::N> t;
#pragma unroll
  for (int k = 0; k < FT::N; k++) t.v[k] = tab[(size_t)idx * FT::STRIDE + k];
  return t;
}

// x * I, I = w^(n/4): a.wq_w = its shifted multiples (ctx.cpp build_wq_w); Ft63: the table entry w^(n/4) R' itself (wave-uniform load)
template <class FT> LCPC_DEV LN<FT::N> mul_i(const LN<FT::N>& x, const NttPassArgs& a) {
  if constexpr (ln::has_mul_u<FT>) return ln::mul_u<FT>(x, a.wq_w);
  else return ln::mul<FT>(x, tab_entry<FT>(a.roots29, 1u << (a.log_n - 2)));
}

// round structure of a pass with S stages on tiles of 2^S x 2^LBT slots (as in ntt_l9s.hip)
template <int S, int LBT> struct Shape {
  static constexpr int U0 = S & 1;
  static constexpr int NR4 = S / 2;
  static constexpr u32 period2 = 1u << (S - 1 + LBT);
  static constexpr u32 period4(int r) { return 1u << (S - U0 - 2 * r - 2 + LBT); }
  // the uniform rounds of ntt_l9s.hip (Shape::RU there): from the radix-4 round whose twiddle period is 4 on, wave w takes the quads
  // q = w mod 4 and all its lanes multiply by the same three twiddles -- scalar operands of ln::mul_u (fields that have one)
  static constexpr int RU = NR4 >= 4 ? 3 : -1;
  static constexpr int NRU = RU < 0 ? 0 : NR4 - RU;
};
// words per shifted-multiples table in the packs: N^2 = 25 / 49 used
template <class FT> constexpr u32 U_SLOT = FT::N == 5 ? 32u : 64u;

constexpr u32 TILE = 1024;

template <class FT, int S, int LTJ, bool FIRST>
__global__ void __launch_bounds__(256, FT::WAVES) ntt_pass_lns_kernel(NttPassArgs a, const u32* __restrict__ pack, NttPackInfo pi) {
  constexpr int N = FT::N, NL = FT::NL, LBT = LTJ;
  constexpr bool LAST = !FIRST;
  static_assert(S + LTJ == 10, "1024-element tiles: one radix-4 quad (two radix-2 pairs) per thread and round");
  static_assert(FIRST || (LTJ == 0 && S % 2 == 0), "the last pass works on contiguous tiles and ends with the trivial stages k-2, k-1");
  using SH = Shape<S, LBT>;
  using E = LN<N>;
  constexpr u32 T = TILE;
  constexpr int RU = ln::has_mul_u<FT> ? SH::RU : -1;
  // passes with a uniform round keep the tile XOR-swizzled, as K1s does (that round reads it with a lane stride of 16 elements)
  auto SWZ = [](u32 e) -> u32 { if constexpr (RU >= 0) return e ^ ((e >> 4) & 15u); else return e; };
  extern __shared__ __attribute__((aligned(16))) u32 lds[];
  u32* nqp = lds + (size_t)T * N;                            // NEGATED q*p rows (ln::clamp_apply)
  const u32 k = a.log_n;
  const u32 tiles_per_row = 1u << (k - 10);
  u64 row;
  u32 tile;
  if (FIRST && a.tile_group) {                               // short runs: neighbouring tiles back to back on one XCD (ntt_l9s.hip)
    const u32 lg = a.tile_group;
    const u32 xcd = blockIdx.x & 7u;
    const u64 qq = blockIdx.x >> 3;
    const u32 sub = (u32)qq & ((1u << lg) - 1);
    const u64 q2 = qq >> lg;
    tile = (((u32)(q2 / a.n_rows) * 8u + xcd) << lg) | sub;
    row = q2 % a.n_rows;
  } else if (tiles_per_row >= 8) {                           // XCD-aware order, as in ntt_pass_kernel
    const u32 xcd = blockIdx.x & 7u;
    const u64 qq = blockIdx.x >> 3;
    tile = (u32)(qq / a.n_rows) * 8u + xcd;
    row = qq % a.n_rows;
  } else {
    row = blockIdx.x / tiles_per_row;
    tile = blockIdx.x % tiles_per_row;
  }
  const u32 tid = threadIdx.x;
  mem_phase(true);                                           // (field_dev.h: wave priority while a wave issues its memory instructions)
  const u32 lb = FIRST ? k - S : 0u;                         // index bits below the pass's stage field
  // element index of LDS slot e = (i << LBT) | lp.  First pass: (i << lb) | (tile << LTJ) | lp; last pass: tile * 2^S + i
  auto gindex = [&](u32 e) -> u32 {
    if constexpr (FIRST) return ((e >> LBT) << lb) | (tile << LTJ) | (e & ((1u << LBT) - 1));
    else return (tile << S) | e;
  };
  for (u32 i = tid; i < 64 * FT::STRIDE; i += 256) nqp[i] = 0u - a.qp29[i];
  const u32* src = a.src + row * a.src_stride * NL;
  // The four elements a thread loads, e = tid + 256 it, ARE the quad (the two pairs) of its first round (dq = 256 there whatever the
  // pass shape): they go to that round in registers -- no LDS round trip, and no barrier, since that round's pure sum is a sum of
  // loads and is normalised, not clamped (the q*p table is first read a round later).  As in K1s (ntt_l9s.hip)
  // DOUT: a last pass stores from its final round's registers (below) -- Ft63 / Ft127; Ft191's 24-byte elements make those stores a 16 + 8
  // byte pair at a lane stride of 96 bytes, which costs more than the LDS round trip saves (2^24: 1.648 -> 1.680 ms, 2^20 +5 %)
  constexpr bool DOUT = N <= 5;
  E xin[4];
#pragma unroll
  for (u32 it = 0; it < 4; it++) {
    const u32 e = tid + 256u * it;
    const u32 g = gindex(e);
    Fe<NL> v;
    if constexpr (FIRST) {        // zero padding, the ragged tail of the caller's vector and the coeffs copy exist here only
      v = (g < a.n_valid && row * a.src_stride + g < a.n_src_total) ? fe_load<NL>(src + (size_t)g * NL) : fe_zero<NL>();
      if (a.copy_dst != nullptr && g < a.n_valid) fe_store<NL>(a.copy_dst + (row * a.src_stride + g) * NL, v);
    } else {
      v = fe_load<NL>(src + (size_t)g * NL);                 // < 2^(32 NL) (the first pass's store), not necessarily < p
    }
    xin[it] = ln::from_packed<FT>(v);
  }
  const u32* cls_pack = pack + (size_t)(FIRST ? tile : 0u) * pi.class_words;
  const bool canon = a.roots29c != nullptr && ((u32)row & a.canon_row_mask) == 0;     // (wave-uniform; three-pass plans: kernels.h)
  // tiles that hold elements of "block 0" (never multiplied so far); a.blk0_gone: an earlier pass with a uniform round converted what
  // was left of it (below)
  const bool blk0_tile = (FIRST || tile == 0) && a.blk0_gone == 0;
  const bool zero_hi = FIRST && a.n_valid <= (1ull << (k - 1));

  if constexpr (SH::U0 == 1) {
    // ---- radix-2 round at stage 0 (odd S): pairs (e1, e1 + half), twiddle w^(index of e1); inputs straight from the loads
    constexpr u32 half = 1u << (S - 1 + LBT);
    const u32* blk = cls_pack + pi.round_off[0];
#pragma unroll
    for (u32 pp = 0; pp < 2; pp++) {
      const u32 e1 = tid + 256u * pp;                        // slots with the top stage bit clear are [0, half)
      const E w = planes_get<FT>(blk, 2 * SH::period2, (canon ? SH::period2 : 0u) + e1);   // stage 0 is all block 0
      const E x = xin[pp];
      if (pp == 0) mem_phase(false);
      if (zero_hi) {
        planes_put<FT>(lds, T, SWZ(e1), x);
        planes_put<FT>(lds, T, SWZ(e1 + half), ln::mul<FT>(x, w));         // (x, 0) -> (x, x w)
      } else {
        const E y = xin[pp + 2];
        E sum = ln::add(x, y);                               // [0, 2p + 128 B)
        ln::normalize<FT>(sum);
        planes_put<FT>(lds, T, SWZ(e1), sum);
        planes_put<FT>(lds, T, SWZ(e1 + half), ln::mul<FT>(ln::sub(x, y), w));
      }
    }
    mem_phase(true);
    __syncthreads();
  }

  const u32 q = tid;                                         // one quad per thread per radix-4 round (T / 4 == 256)
#pragma unroll
  for (int r = 0; r < SH::NR4; r++) {
    const int u = SH::U0 + 2 * r, hb = S - u - 1;            // stages (u, u + 1) of this pass; pair bit of stage u
    const bool last_two = LAST && (r == SH::NR4 - 1);        // stages k-2, k-1: twiddles 1, w^(n/4), 1
    const u32 lp = q & ((1u << LBT) - 1), j = q >> LBT;
    const u32 i0 = ((j >> (hb - 1)) << (hb + 1)) | (j & ((1u << (hb - 1)) - 1));
    const u32 e0 = (i0 << LBT) | lp;
    constexpr u32 one = 1u;
    const u32 dq = one << (hb - 1 + LBT);
    const u32 period = one << (hb - 1 + LBT);                // quads q and q + period share their twiddles
    const u32 jl = q & (period - 1);
    const u32* blk = cls_pack + pi.round_off[SH::U0 + r];
    if constexpr (RU >= 0) {
      if (r >= RU && !last_two) {
        // ---- a uniform round (ntt_l9s.hip): wave w takes the quads q = w mod 4; its three twiddles are scalar operands.  Block 0 is
        //      gone (converted in round RU - 1), so every lane multiplies by the same plain constants
        const u32 wv = __builtin_amdgcn_readfirstlane(tid >> 6);
        const u32 qu = ((tid & 63u) << 2) | wv;
        const u32 lpu = qu & ((1u << LBT) - 1), ju = qu >> LBT;
        const u32 iu = ((ju >> (hb - 1)) << (hb + 1)) | (ju & ((1u << (hb - 1)) - 1));
        const u32 eu = (iu << LBT) | lpu;
        const u32* wu = cls_pack + pi.u_off + ((r - RU) * 4 + wv) * (3 * U_SLOT<FT>);
        const E x0 = planes_get<FT>(lds, T, SWZ(eu)), x1 = planes_get<FT>(lds, T, SWZ(eu + dq));
        const E x2 = planes_get<FT>(lds, T, SWZ(eu + 2 * dq)), x3 = planes_get<FT>(lds, T, SWZ(eu + 3 * dq));
        mem_phase(false);
        const E b0 = ln::add(x0, x2), b1 = ln::add(x1, x3);
        E c0 = ln::add(b0, b1);
        ln::clamp_apply<FT>(c0, ln::clamp_row<FT>(nqp, ln::clamp_q<FT>(c0.v[N - 1])));
        planes_put<FT>(lds, T, SWZ(eu), c0);
        E d1 = ln::sub(b0, b1);
        ln::normalize<FT>(d1);                                                                      // mul_u wants sum |limb| < N 2^W
        planes_put<FT>(lds, T, SWZ(eu + dq), ln::mul_u<FT>(d1, wu + 2 * U_SLOT<FT>));
        const E b2 = ln::mul_u<FT>(ln::sub(x0, x2), wu);
        const E b3 = ln::mul_u<FT>(ln::sub(x1, x3), wu + U_SLOT<FT>);                               // (-2.5p, 1.6p)
        E c2 = ln::add(b2, b3);
        ln::normalize<FT>(c2);
        planes_put<FT>(lds, T, SWZ(eu + 2 * dq), c2);
        planes_put<FT>(lds, T, SWZ(eu + 3 * dq), ln::mul_u<FT>(ln::sub(b2, b3), wu + 2 * U_SLOT<FT>));
        mem_phase(true);
        __syncthreads();
        continue;
      }
    }
    if (u == 0 && zero_hi) {
      // zero-padded first round (rate <= 1/2): x2 = x3 = 0, the stage-0 butterflies are (x, x w); inputs < 4p; everything
      // is block 0, so with canonical output the three multiplies leaving it (w0, w3, and w2 for c1) take the converting set
      const u32 vb = canon ? 3u : 0u;
      const E w0 = planes_get<FT>(blk, 6 * period, (vb + 0) * period + jl), w2 = planes_get<FT>(blk, 6 * period, (vb + 2) * period + jl);
      const E w3 = planes_get<FT>(blk, 6 * period, (vb + 1) * period + jl);
      mem_phase(false);
      if (a.n_valid <= (1ull << (k - 2))) {                  // rate <= 1/4: x1 is zero too
        const E x0 = xin[0];
        planes_put<FT>(lds, T, SWZ(e0), x0);
        planes_put<FT>(lds, T, SWZ(e0 + dq), ln::mul<FT>(x0, w2));
        planes_put<FT>(lds, T, SWZ(e0 + 2 * dq), ln::mul<FT>(x0, w0));
        planes_put<FT>(lds, T, SWZ(e0 + 3 * dq), ln::mul<FT>(x0, w3));
      } else {
        const E x0 = xin[0], x1 = xin[1];
        E c0 = ln::add(x0, x1);
        ln::normalize<FT>(c0);
        planes_put<FT>(lds, T, SWZ(e0), c0);
        planes_put<FT>(lds, T, SWZ(e0 + dq), ln::mul<FT>(ln::sub(x0, x1), w2));
        const E t = mul_i<FT>(x1, a);                                                        // x1 I: keeps x1's form; normalised, (-2.5p, 1.6p)
        planes_put<FT>(lds, T, SWZ(e0 + 2 * dq), ln::mul<FT>(ln::add(x0, t), w0));
        planes_put<FT>(lds, T, SWZ(e0 + 3 * dq), ln::mul<FT>(ln::sub(x0, t), w3));
      }
      mem_phase(true);
      __syncthreads();
      continue;
    }
    const bool from_regs = SH::U0 == 0 && r == 0;             // (e0 == tid, dq == 256: the thread's own loads)
    const E x0 = from_regs ? xin[0] : planes_get<FT>(lds, T, SWZ(e0)), x1 = from_regs ? xin[1] : planes_get<FT>(lds, T, SWZ(e0 + dq));
    const E x2 = from_regs ? xin[2] : planes_get<FT>(lds, T, SWZ(e0 + 2 * dq)), x3 = from_regs ? xin[3] : planes_get<FT>(lds, T, SWZ(e0 + 3 * dq));   // I: normalised, |value| < 4p
    mem_phase(false);
    const E b0 = ln::add(x0, x2), b1 = ln::add(x1, x3);                                           // limbs [0, 2^(W+1)), |value| < 8p
    E c0 = ln::add(b0, b1);                                                                       // limbs [0, 2^(W+2)), |value| < 16p
    if (last_two || from_regs) ln::normalize<FT>(c0);                                             // (the store path clamps every slot; sums of loads < 4.01 p)
    else ln::clamp_apply<FT>(c0, ln::clamp_row<FT>(nqp, ln::clamp_q<FT>(c0.v[N - 1])));           // [0, p + 64 B)
    if (!(last_two && DOUT)) planes_put<FT>(lds, T, SWZ(e0), c0);
    if (last_two) {
      // outputs go straight to the store (normalised, |value| < 16p)
      E c1 = ln::sub(b0, b1);
      const E b2 = ln::sub(x0, x2);
      const E b3 = mul_i<FT>(ln::sub(x1, x3), a);                                                 // w^(n/4): the one twiddle every lane shares
      E c2 = ln::add(b2, b3);
      E c3 = ln::sub(b2, b3);
      ln::normalize<FT>(c1); ln::normalize<FT>(c2); ln::normalize<FT>(c3);
      // dq == 1 here: the quad is four CONSECUTIVE elements of the row -- reduced and stored from the registers (no LDS round trip, no
      // barrier; ntt_l9s.hip).  -> [0, p): after the clamp, value >= p needs the top limb to reach floor(p / B)
      if constexpr (!DOUT) {
        planes_put<FT>(lds, T, SWZ(e0 + dq), c1);
        planes_put<FT>(lds, T, SWZ(e0 + 2 * dq), c2);
        planes_put<FT>(lds, T, SWZ(e0 + 3 * dq), c3);
      } else {
      mem_phase(true);
      u32* dstq = a.dst + row * a.dst_stride * NL + (size_t)((tile << S) | e0) * NL;
      E cc[4] = {c0, c1, c2, c3};
#pragma unroll
      for (int c = 0; c < 4; c++) {
        E x = cc[c];
        ln::clamp_apply<FT>(x, ln::clamp_row<FT>(nqp, ln::clamp_q<FT>(x.v[N - 1])));  // [0, p + 64 B) < 2^(32 NL)
        u32 w[NL];
        ln::to_packed<FT>(w, x.v);
        Fe<NL> v;
        if (__any((int)(x.v[N - 1] >= FT::limb(N - 1)))) v = fe_reduce_once<NL>(w, 0u);
        else {
#pragma unroll
          for (int i = 0; i < NL; i++) v.v[i] = w[i];
        }
        if (canon && tile == 0 && e0 + (u32)c < a.mont_prefix) v = fe_canon<NL>(v);     // canonical output: the never-multiplied prefix (Ft63)
        fe_store<NL>(dstq + (size_t)c * NL, v);
      }
      return;
      }
    } else {
      // block 0 of stages (u, u + 1) = the quads whose elements all lie below n / 2^(t + 2): exactly q < period in the tiles
      // that hold block 0.  Their three multiplies that leave block 0 (c1, c2, c3) take the converting set; c0 stays a pure sum;
      // t = (x1 - x3) I is by a plain constant and stays in the form of its inputs
      const bool blk0c = canon && blk0_tile && q < period && (RU < 0 || r < RU);
      if constexpr (RU >= 1) {
        if (r == RU - 1 && blk0c) {
          // the last round before the uniform one: c0, the pure sum that would carry block 0 on, is converted as well -- a multiply by
          // R' R^-1 = 2^(N W - 32 NL) (16 lanes of one wave per block-0 tile) -- so that the uniform round sees canonical values only
          E kc;
#pragma unroll
          for (int i = 0; i < N; i++) kc.v[i] = i == 0 ? (1u << (N * FT::W - 32 * NL)) : 0u;
          planes_put<FT>(lds, T, SWZ(e0), ln::mul<FT>(c0, kc));
        }
      }
      const u32 vb = blk0c ? 3u : 0u;
      const E w2 = planes_get<FT>(blk, 6 * period, (vb + 2) * period + jl);
      planes_put<FT>(lds, T, SWZ(e0 + dq), ln::mul<FT>(ln::sub(b0, b1), w2));                          // in: |value| < 16p
      const E t = mul_i<FT>(ln::sub(x1, x3), a);                                                  // normalised, (-2.5p, 1.6p)
      const E e2 = ln::sub(x0, x2);                                                               // limbs (-2^W, 2^W), |value| < 8p
      const E w0 = planes_get<FT>(blk, 6 * period, (vb + 0) * period + jl), w3 = planes_get<FT>(blk, 6 * period, (vb + 1) * period + jl);
      planes_put<FT>(lds, T, SWZ(e0 + 2 * dq), ln::mul<FT>(ln::add(e2, t), w0));                       // in: limbs (-2^W, 2^(W+1)), |value| < 10.5p
      planes_put<FT>(lds, T, SWZ(e0 + 3 * dq), ln::mul<FT>(ln::sub(e2, t), w3));
    }
    mem_phase(true);
    __syncthreads();
  }

  if constexpr (FIRST || N > 5) {           // (a last pass with DOUT has stored from its final round and returned)
  u32* dst = a.dst + row * a.dst_stride * NL;
#pragma unroll
  for (u32 e = tid; e < T; e += 256) {
    const u32 g = gindex(e);
    E x = planes_get<FT>(lds, T, SWZ(e));                                            // normalised, |value| < 16p
    ln::clamp_apply<FT>(x, ln::clamp_row<FT>(nqp, ln::clamp_q<FT>(x.v[N - 1])));  // [0, p + 64 B) < 2^(32 NL): the successor reads limbs anyway
    u32 w[NL];
    ln::to_packed<FT>(w, x.v);
    Fe<NL> v;
#pragma unroll
    for (int i = 0; i < NL; i++) v.v[i] = w[i];
    if constexpr (LAST) {
      // -> [0, p): after the clamp, value >= p needs the top limb to reach floor(p / B)
      if (__any((int)(x.v[N - 1] >= FT::limb(N - 1)))) v = fe_reduce_once<NL>(w, 0u);
      if (canon && tile == 0 && g < a.mont_prefix) v = fe_canon<NL>(v); // canonical output: the never-multiplied prefix
    }
    fe_store<NL>(dst + (size_t)g * NL, v);
  }
  }
}

// one thread per (class, round slot, position): copies the table entries a quad / pair will ask for into lane order
template <class FT, int S, int LBT>
__global__ void __launch_bounds__(256) ntt_lns_pack_kernel(NttPassArgs a, NttPackInfo pi, u32 n_classes, bool first, u32* pack) {
  using SH = Shape<S, LBT>;
  constexpr u32 n_slots = SH::U0 + SH::NR4;
  constexpr u32 PMAX = 1u << (S - 1 + LBT);                  // >= every period
  const u32 k = a.log_n, t0 = a.t0;
  const u32 lb = first ? k - S : 0u;
  const u64 total = (u64)n_classes * n_slots * PMAX;
  for (u64 id = (u64)blockIdx.x * 256 + threadIdx.x; id < total; id += (u64)gridDim.x * 256) {
    const u32 jl = (u32)(id % PMAX), slot = (u32)((id / PMAX) % n_slots), cls = (u32)(id / PMAX / n_slots);
    u32* blk = pack + (size_t)cls * pi.class_words + pi.round_off[slot];
    const u32 lo = first ? (cls << LBT) : 0u;                // first pass: the tile's own low index bits
    if (SH::U0 == 1 && slot == 0) {                          // radix-2 round at stage t0: w^((g1 & gm) << t0), g1 = index of e1 = jl
      const u32 lp = jl & ((1u << LBT) - 1), i = jl >> LBT;
      const u32 g1 = (i << lb) | lo | lp;
      const u32 gm = (1u << (k - t0 - 1)) - 1;
      planes_put<FT>(blk, 2 * SH::period2, jl, tab_entry<FT>(a.roots29, (g1 & gm) << t0));
      planes_put<FT>(blk, 2 * SH::period2, SH::period2 + jl, tab_entry<FT>(a.roots29c, (g1 & gm) << t0));
      continue;
    }
    const u32 r = slot - SH::U0, u = SH::U0 + 2 * r, hb = S - u - 1, period = 1u << (hb - 1 + LBT);
    const u32 t = t0 + u;
    if (jl >= period || t + 2 == k) continue;                // (stages k-2, k-1: one wave-uniform twiddle, not packed)
    const u32 gm0 = (1u << (k - t - 1)) - 1;
    const u32 lp = jl & ((1u << LBT) - 1), j = jl >> LBT;
    const u32 i0 = ((j >> (hb - 1)) << (hb + 1)) | (j & ((1u << (hb - 1)) - 1));
    const u32 g0 = (i0 << lb) | lo | lp;                     // (last pass: the tile's high bits do not reach these twiddles)
    // w0 = w^e, w3 = w^(3 e), w2 = w^(2 e), e = (g0 & gm0) << t < n / 4; the tables hold w^i for i < n / 2 and w^(n/2) = -1: past
    // that, the negated entry (ntt_l9s.hip ntt_pack_kernel)
    const u32 ex = (g0 & gm0) << t, half_n = 1u << (k - 1);
    const u32 idx[3] = {ex, 3 * ex, 2 * ex};
    for (u32 v = 0; v < 6; v++) {
      const u32 ix = idx[v % 3];
      LN<FT::N> m = tab_entry<FT>(v < 3 ? a.roots29 : a.roots29c, ix & (half_n - 1));
      if (ix >= half_n) {                                    // p - entry, limb-wise with borrow (entry in (0, p))
        int32_t br = 0;
#pragma unroll
        for (int z = 0; z < FT::N; z++) {
          const int32_t d = (int32_t)FT::limb(z) - (int32_t)m.v[z] - br;
          br = d < 0 ? 1 : 0;
          m.v[z] = z + 1 < FT::N ? (u32)d & ((1u << FT::W) - 1) : (u32)d;
        }
      }
      planes_put<FT>(blk, 6 * period, v * period + jl, m);
    }
  }
}

// the uniform rounds' constants (ntt_l9s.hip ntt_upack_kernel for N limbs of W bits): per class, for jl = 0..3 and the round's three
// twiddles w0, w1 = I w0, w2 (plain: block 0 is gone by then), the N shifted multiples W_j = balanced(w 2^(W j) mod p) as N^2 words
// t = N k + j (limb k of W_j; ln::mul_u / field_wmul_gen.h).  The table entry is w R' mod p: ln::mul(2^(W j), entry) = w 2^(W j),
// lazily reduced in (-p - eps, eps]; + p where that lies below -(p - 1) / 2.
template <class FT, int S, int LBT>
__global__ void __launch_bounds__(64) ntt_lns_upack_kernel(NttPassArgs a, NttPackInfo pi, u32 n_classes, bool first, u32* pack) {
  using SH = Shape<S, LBT>;
  constexpr int N = FT::N, W = FT::W;
  constexpr u32 M = (1u << W) - 1;
  const u32 k = a.log_n, t0 = a.t0;
  const u32 lb = first ? k - S : 0u;
  const u32 id = blockIdx.x * 64 + threadIdx.x;
  if (id >= n_classes * SH::NRU * 12) return;
  const u32 cls = id / (SH::NRU * 12), ru = (id / 12) % SH::NRU, jl0 = (id % 12) / 3, v = id % 3;
  const u32 r = SH::RU + ru;
  const u32 u = SH::U0 + 2 * r, hb = S - u - 1;
  const u32 t = t0 + u;
  if (t + 2 == k) return;                                    // (a last pass's final round: w^(n/4), not packed)
  const u32 jl = jl0 & (SH::period4(r) - 1);                 // periods 2 and 1: the four slots repeat
  const u32 lo = first ? (cls << LBT) : 0u;
  const u32 gm0 = (1u << (k - t - 1)) - 1, gm1 = gm0 >> 1;
  const u32 lp = jl & ((1u << LBT) - 1), j = jl >> LBT;
  const u32 i0 = ((j >> (hb - 1)) << (hb + 1)) | (j & ((1u << (hb - 1)) - 1));
  const u32 g0 = (i0 << lb) | lo | lp;
  const u32 g1 = g0 + (1u << (hb - 1 + lb));
  const u32 idx = v == 0 ? (g0 & gm0) << t : (v == 1 ? (g1 & gm0) << t : (g0 & gm1) << (t + 1));
  const LN<N> w = tab_entry<FT>(a.roots29, idx);
  u32* out = pack + (size_t)cls * pi.class_words + pi.u_off + ((ru * 4 + jl0) * 3 + v) * U_SLOT<FT>;
  for (u32 jj = 0; jj < (u32)N; jj++) {
    LN<N> sh;
#pragma unroll
    for (int z = 0; z < N; z++) sh.v[z] = (u32)z == jj ? 1u : 0u;
    LN<N> x = ln::mul<FT>(sh, w);                            // w 2^(W jj) mod p, in (-p - eps, eps], normalised
    LN<N> tt;                                                // x + (p - 1) / 2 < 0  <=>  x below the balanced range
#pragma unroll
    for (int z = 0; z < N; z++) {
      const u32 hz = z + 1 < N ? ((FT::limb(z) >> 1) | ((FT::limb(z + 1) & 1u) << (W - 1))) : (FT::limb(z) >> 1);
      tt.v[z] = x.v[z] + hz;
    }
    ln::normalize<FT>(tt);
    if ((int32_t)tt.v[N - 1] < 0) {
#pragma unroll
      for (int z = 0; z < N; z++) x.v[z] += FT::limb(z);
      ln::normalize<FT>(x);
    }
#pragma unroll
    for (int kk = 0; kk < N; kk++) out[N * kk + jj] = kk + 1 < N ? (x.v[kk] & M) : x.v[kk];
  }
  for (u32 z = N * N; z < U_SLOT<FT>; z++) out[z] = 0;
}

template <class FT, int S, int LBT> NttPackInfo pack_info_t() {
  using SH = Shape<S, LBT>;
  NttPackInfo pi{};
  u32 off = 0, slot = 0;
  // radix-2 slot: variants (plain, converting); radix-4 slots: w0, w3 = w0 w2, w2 plain, then the same from the converting table
  if (SH::U0) { pi.round_off[slot++] = off; off += 2 * SH::period2 * FT::N; off = (off + 3) & ~3u; }
  for (int r = 0; r < SH::NR4; r++) { pi.round_off[slot++] = off; off += 6 * SH::period4(r) * FT::N; off = (off + 3) & ~3u; }
  if (ln::has_mul_u<FT> && SH::RU >= 0) { off = (off + 15) & ~15u; pi.u_off = off; off += SH::NRU * 4 * 3 * U_SLOT<FT>; }
  pi.class_words = off;
  return pi;
}

template <class FT, int S, int LTJ, bool FIRST>
hipError_t launch_t(const NttPassArgs& a, const u32* pack, const NttPackInfo& pi, hipStream_t st) {
  const u64 tiles = ((u64)1 << (a.log_n - 10)) * a.n_rows;
  const size_t lds_bytes = ((size_t)TILE * FT::N + 64 * FT::STRIDE) * 4;
  hipLaunchKernelGGL((ntt_pass_lns_kernel<FT, S, LTJ, FIRST>), dim3((unsigned)tiles), dim3(256), lds_bytes, st, a, pack, pi);
  return hipGetLastError();
}

// limb-form twiddle table: out[i] = roots[i] * (R' mod p) / R = w^i R' mod p, fully reduced, as N limbs of W bits
template <class FT>
__global__ void __launch_bounds__(256) roots_ln_kernel(const u32* roots, u64 n, const u32* rprime, u32* out) {
  const u64 i = (u64)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const Fe<FT::NL> c = fe_load<FT::NL>(rprime);
  const Fe<FT::NL> t = fe_mul<FT::NL>(fe_load<FT::NL>(roots + i * FT::NL), c);
  const LN<FT::N> l = ln::from_packed<FT>(t);
#pragma unroll
  for (int k = 0; k < FT::STRIDE; k++) out[i * FT::STRIDE + k] = k < FT::N ? l.v[k] : 0u;
}

#define LNS_FIRST_CASES(X) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10)

template <class FT> NttPackInfo pack_info_f(uint32_t s, bool first) {
  if (!first) return pack_info_t<FT, 10, 0>();
  switch (s) {
#define X(SV) case SV: return pack_info_t<FT, SV, 10 - SV>();
    LNS_FIRST_CASES(X)
#undef X
  }
  return NttPackInfo{};
}
template <class FT> hipError_t launch_pack_f(const NttPassArgs& a, bool first, const NttPackInfo& pi, uint32_t n_classes, uint32_t* pack, hipStream_t st) {
  const unsigned ugrid = (n_classes * 2 * 12 + 63) / 64;      // (<= 2 uniform rounds per pass)
  if (!first) {
    hipLaunchKernelGGL((ntt_lns_pack_kernel<FT, 10, 0>), dim3(64), dim3(256), 0, st, a, pi, n_classes, false, pack);
    if constexpr (ln::has_mul_u<FT>) hipLaunchKernelGGL((ntt_lns_upack_kernel<FT, 10, 0>), dim3(ugrid), dim3(64), 0, st, a, pi, n_classes, false, pack);
    return hipGetLastError();
  }
  switch (a.s) {
#define X(SV) case SV: hipLaunchKernelGGL((ntt_lns_pack_kernel<FT, SV, 10 - SV>), dim3(2048), dim3(256), 0, st, a, pi, n_classes, true, pack); break;
    LNS_FIRST_CASES(X)
#undef X
    default: return hipErrorInvalidValue;
  }
  if constexpr (ln::has_mul_u<FT>) {
    if (a.s == 8) hipLaunchKernelGGL((ntt_lns_upack_kernel<FT, 8, 2>), dim3(ugrid), dim3(64), 0, st, a, pi, n_classes, true, pack);
    if (a.s == 9) hipLaunchKernelGGL((ntt_lns_upack_kernel<FT, 9, 1>), dim3(ugrid), dim3(64), 0, st, a, pi, n_classes, true, pack);
    if (a.s == 10) hipLaunchKernelGGL((ntt_lns_upack_kernel<FT, 10, 0>), dim3(ugrid), dim3(64), 0, st, a, pi, n_classes, true, pack);
  }
  return hipGetLastError();
}
template <class FT> hipError_t launch_pass_f(const NttPassArgs& a, bool first, const uint32_t* pack, const NttPackInfo& pi, hipStream_t st) {
  if (!first) {
    if (a.s != 10 || a.log_tj != 0 || a.t0 + a.s != a.log_n) return hipErrorInvalidValue;
    return launch_t<FT, 10, 0, false>(a, pack, pi, st);
  }
  if (a.t0 != 0 || a.s + a.log_tj != 10 || (a.s + 10 != a.log_n && a.s + 20 != a.log_n)) return hipErrorInvalidValue;   // (+ 20: three-pass plans)
  switch (a.s) {
#define X(SV) case SV: return launch_t<FT, SV, 10 - SV, true>(a, pack, pi, st);
    LNS_FIRST_CASES(X)
#undef X
  }
  return hipErrorInvalidValue;
}

}  // namespace

#define LNS_DISPATCH(nl, EXPR)                         \
  switch (nl) {                                        \
    case 2: { using FT = LnField<FT63>; EXPR; }        \
    case 4: { using FT = LnField<FT127>; EXPR; }       \
    case 6: { using FT = LnField<FT191>; EXPR; }       \
    default: break;                                    \
  }

// n_cols up to 2^20: measured against the general kernel's three-pass plans (round 3, 2^25-element matrices): 2^19 columns
// 0.73 / 1.43 / 2.30 ms against 0.76 / 1.63 / 2.90; 2^20 columns 1.97 / 2.26 ms against 2.02 / 3.26 for Ft127 / Ft191.  Ft63's
// 8-byte first-pass runs lost at 2^20 columns until the first pass ran the tiles that share cache lines back to back on one
// XCD (NttPassArgs.tile_group): 512 rows x 2^20 columns 12.9 ms on the general plan, 23.3 ungrouped, 11.2 grouped.
bool ntt_lns_supported(int nl, uint32_t log_n) { return (nl == 2 || nl == 4 || nl == 6) && log_n >= 11 && log_n <= 20u; }
bool ntt_lns3_supported(int nl, uint32_t log_n) { return (nl == 2 || nl == 4 || nl == 6) && log_n >= 21 && log_n <= 26; }
__global__ void __launch_bounds__(256) lns_subtable_kernel(const u32* tab, u32 shift, u64 n, u32 words, u32* sub) {
  for (u64 id = (u64)blockIdx.x * 256 + threadIdx.x; id < n * words; id += (u64)gridDim.x * 256) {
    const u64 i = id / words, w = id % words;
    sub[i * words + w] = tab[(i << shift) * words + w];
  }
}
hipError_t launch_ntt_lns_subtable(int nl, const uint32_t* tab, uint32_t shift, uint64_t n, uint32_t* sub, hipStream_t st) {
  hipLaunchKernelGGL(lns_subtable_kernel, dim3(2048), dim3(256), 0, st, tab, shift, n, (u32)ntt_lns_stride(nl), sub);
  return hipGetLastError();
}
int ntt_lns_limbs(int nl) { return nl == 2 ? 3 : (nl == 4 ? 5 : (nl == 6 ? 7 : 0)); }
int ntt_lns_limb_bits(int nl) { return nl == 2 ? 26 : 29; }
int ntt_lns_stride(int nl) { return nl == 2 ? 4 : 8; }

NttPackInfo ntt_lns_pack_info(int nl, uint32_t s, bool first) {
  LNS_DISPATCH(nl, return pack_info_f<FT>(s, first))
  return NttPackInfo{};
}
hipError_t launch_ntt_lns_roots(int nl, const uint32_t* roots, uint64_t n, const uint32_t* rprime, uint32_t* out, hipStream_t st) {
  const dim3 grid((unsigned)((n + 255) / 256));
  LNS_DISPATCH(nl, hipLaunchKernelGGL((roots_ln_kernel<FT>), grid, dim3(256), 0, st, roots, n, rprime, out); return hipGetLastError())
  return hipErrorInvalidValue;
}
hipError_t launch_ntt_lns_pack(int nl, const NttPassArgs& a, bool first, const NttPackInfo& pi, uint32_t n_classes, uint32_t* pack, hipStream_t st) {
  LNS_DISPATCH(nl, return launch_pack_f<FT>(a, first, pi, n_classes, pack, st))
  return hipErrorInvalidValue;
}
hipError_t launch_ntt_pass_lns(int nl, const NttPassArgs& a, bool first, const uint32_t* pack, const NttPackInfo& pi, hipStream_t st) {
  if (a.tile_group && (!first || ((1u << (a.log_n - 10)) >> a.tile_group) < 8)) return hipErrorInvalidValue;
  LNS_DISPATCH(nl, return launch_pass_f<FT>(a, first, pack, pi, st))
  return hipErrorInvalidValue;
}

}  // namespace lcpc
