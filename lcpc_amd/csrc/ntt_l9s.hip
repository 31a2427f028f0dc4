// lcpc_amd/csrc/ntt_l9s.hip -- K1s: the shape-specialised Ft255 row NTT (LcEncoding::encode for Ligero,
// lcpc-ligero-pc/src/lib.rs:162-164 = fffft fft_io_pc [3P]) for the two-pass plans on 1024-element tiles
// (2^11 <= n_cols <= 2^20, which covers BASELINE.json's Ligero configs) and, built from the same two kernels, three-pass plans
// for 2^21 .. 2^26 columns (kernels.h ntt_l9s3_supported).
//
// Same tiling, same lazy signed 9 x 29-bit arithmetic and the same canonical-output trick as ntt_pass_l9_kernel
// (kernels.hip, which stays the general kernel: one pass, three passes, 2048-element tiles); what differs is where the
// non-multiplier instructions went (profiles/r02_ntt_lab_*.txt: the kernel is VALU-issue bound, every instruction counts):
//   * the pass shape -- S stages on 2^S x 2^LTJ tiles, first or last pass -- is a template parameter, so the index
//     math of a round is a handful of shifts instead of ~40 instructions on run-time shifts and masks;
//   * twiddles come from a per-pass PACK in lane order: for tile class c (first pass: the tile's position in the row;
//     last pass: one class) and round r the twiddles of quad q sit at [c][r][variant][chunk][q mod period], so a wave
//     reads 1 KiB runs instead of gathering 64 x 48-byte table entries at strides of up to 12 KiB;
//   * normalise + clamp of the pure-sum output are one carry pass (l9::clamp_apply), done before the multiplier
//     chains start so that the sum does not occupy registers across them (no scratch spills at 128 VGPRs);
//   * the first pass stores values in [0, p + 2^239) (< 2^256, what its successor reads as limbs anyway) without the
//     final conditional subtract; the last pass does that subtract only for the rare waves that need it;
//   * an odd stage count is peeled as a radix-2 round at stage 0, where a zero-padded row (rate <= 1/2) needs no
//     additions at all: (x, 0) -> (x, x w).
//   * (round 6) a tile makes no LDS round trip it does not need: the four elements a thread loads are its first round's quad, which
//     therefore runs from the registers with no barrier before it; the last pass's final round owns four consecutive elements and
//     reduces and stores them itself;
//   * (round 6) a wave holds priority 1 while it issues memory instructions and 0 inside the multiplier chains (field_dev.h mem_phase).
// Exact modular arithmetic: any stage grouping gives the same fully-reduced bits as the reference's radix-2 loop.
#include "kernels.h"
#include "ntt_l9_dev.h"

namespace lcpc {

namespace {

// pack block of one (class, round): [NV variants][2 chunks of 16 B][period] uint4, then [NV][period] u32 (limb 8).
// radix-4 rounds: variants 0 w0, 1 w3, 2 w2 (plain table, w^i * 2^261) and 3, 4, 5 the same from the converting table
// (w^i * 2^5); the radix-2 round: 0 w, 1 converting w.
//
// The radix-4 butterfly of stages (t, t + 1) on x0 .. x3 (quarter-block apart), radix-2 DIF regrouped:
//   b0 = x0 + x2, b1 = x1 + x3, b2 = (x0 - x2) w0, b3 = (x1 - x3) w1;  c0 = b0 + b1, c1 = (b0 - b1) w2, c2 = b2 + b3, c3 = (b2 - b3) w2
// with w0 = w^e, w1 = w^(e + n/4) = I w0 (I = w^(n/4), the field's fixed primitive 4th root of unity) and w2 = w0^2.  Hence
//   t = (x1 - x3) I;  c2 = ((x0 - x2) + t) w0;  c3 = ((x0 - x2) - t) w3,  w3 = w0 w2 = w^(3 e):
// I is ONE constant for every lane of every transform, so its multiply takes the shifted-multiples form with scalar operands
// (l9::mul_u on a.wq_w: 119 instructions) and a generic round does three lane-varying Montgomery multiplies (188 each) instead
// of four; c2 comes out of a multiplier normalised.  Exact arithmetic mod p: the same fully reduced bits.
template <u32 NV> LCPC_DEV Fe29 pk_load(const u32* blk, u32 period, u32 variant, u32 jl) {
  // 32-bit byte offsets from the (wave-uniform) block pointer: scalar base + vector offset addressing, no 64-bit VALU adds
  // (a class block is < 2^20 words)
  const char* base = reinterpret_cast<const char*>(blk);
  const uint4 a = *reinterpret_cast<const uint4*>(base + (((variant * 2 + 0) * period + jl) << 4));
  const uint4 b = *reinterpret_cast<const uint4*>(base + (((variant * 2 + 1) * period + jl) << 4));
  const u32 c = *reinterpret_cast<const u32*>(base + ((NV * 2 * period * 4 + variant * period + jl) << 2));
  Fe29 t;
  t.v[0] = a.x; t.v[1] = a.y; t.v[2] = a.z; t.v[3] = a.w; t.v[4] = b.x; t.v[5] = b.y; t.v[6] = b.z; t.v[7] = b.w; t.v[8] = c;
  return t;
}

// round structure of a pass with S stages on tiles of 2^S x 2^LBT slots: an odd S peels stage 0 as a radix-2 round
// (slot 0 of the pack), the radix-4 rounds r = 0, 1, .. then cover stages (U0 + 2r, U0 + 2r + 1)
template <int S, int LBT> struct Shape {
  static constexpr int U0 = S & 1;
  static constexpr int NR4 = S / 2;
  static constexpr u32 period2 = 1u << (S - 1 + LBT);                     // radix-2 round: all 512 pairs differ
  static constexpr u32 period4(int r) { return 1u << (S - U0 - 2 * r - 2 + LBT); }
  // the radix-4 round whose twiddle period is 4 (quads q and q + 4 share their twiddles): with the lanes dealt so that wave w holds
  // the quads q = w mod 4, every lane of a wave multiplies by the SAME three twiddles -- the shifted-multiples multiply with scalar
  // operands (l9::mul_u, 119 instructions against 188).  The same deal serves periods 2 and 1.  S + LBT == 10: the period
  // is 2^(8 - U0 - 2 r), i.e. <= 4 from round 3 on: RU = 3 where the pass has four radix-4 rounds (S >= 8), -1: none.  NRU: how
  // many rounds from RU on (a last pass ends with the trivial stages k-2, k-1, which have their own form).
  static constexpr int RU = NR4 >= 4 ? 3 : -1;
  static constexpr int NRU = RU < 0 ? 0 : NR4 - RU;
};
constexpr u32 U_SLOT = 96;                                  // words per shifted-multiples table (81 used)

// MID: the intermediate between the two passes is not the packed comm buffer but a.mid, which holds every element as the
// LDS tile holds it -- 9 signed 29-bit limbs, normalised, |value| < 4p (invariant I of field_dev.h) -- laid out per row as
// the successor's tiles: [tile of 1024 elements][limbs 0-3 x 1024 | limbs 4-7 x 1024 | limb 8 x 1024] (36 KiB per tile).
// The first pass then stores its tile as it stands (no clamp, no reduction, no 29 -> 32-bit packing: ~60 VALU per element)
// and the last pass's tile load is a plain 36 KiB copy into LDS (no unpacking: ~17 per element), at 36 instead of 32 bytes
// per element of intermediate traffic on a kernel that HBM does not bind.
template <int S, int LTJ, bool FIRST, bool MID>
__global__ void __launch_bounds__(256, 4) ntt_pass_l9s_kernel(NttPassArgs a, const u32* __restrict__ pack, NttPackInfo pi) {
  constexpr int NL = 8, LT = S + LTJ, LBT = LTJ;
  constexpr bool LAST = !FIRST;
  static_assert(LT == 10, "1024-element tiles: one radix-4 quad (two radix-2 pairs) per thread and round");
  static_assert(FIRST || (LTJ == 0 && S % 2 == 0), "the last pass works on contiguous tiles and ends with the trivial stages k-2, k-1");
  using SH = Shape<S, LBT>;
  constexpr int RU = SH::RU;
  // a pass with a uniform round reads its tile with a lane stride of 16 elements there (64-way bank conflicts on the uint4
  // planes): the tile is kept XOR-swizzled -- element e at slot e ^ ((e >> 4) & 15), a permutation inside every aligned block
  // of 16 -- which leaves the consecutive accesses of the other rounds conflict-free and spreads that round's over all banks
  auto SWZ = [](u32 e) -> u32 { if constexpr (RU >= 0) return e ^ ((e >> 4) & 15u); else return e; };
  extern __shared__ __attribute__((aligned(16))) u32 lds[];
  u32* nqp = lds + (size_t)Lds9<LT>::T * 9;                  // NEGATED q*p rows (l9::clamp_apply)
  const u32 k = a.log_n;
  constexpr u32 T = 1u << LT;
  const u32 tiles_per_row = 1u << (k - LT);
  u64 row;
  u32 tile;
  if (FIRST && a.tile_group) {
    // short runs: 2^tile_group neighbouring tiles touch the same 128-byte lines / DRAM pages.  They go to one XCD (workgroups
    // are dealt to the XCDs round-robin) as consecutive workgroups, so that a line one of them fetched is in that L2 when the
    // others ask for it and their partial-line stores meet there before the write-back (host: ntt_tile_group, ctx.cpp)
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
  const bool canon = a.roots29c != nullptr && ((u32)row & a.canon_row_mask) == 0;     // (wave-uniform)
  for (u32 i = tid; i < 64 * 12; i += 256) nqp[i] = 0u - a.qp29[i];
  const u32* src = a.src + row * a.src_stride * NL;
  constexpr bool DIRECT = !(LAST && MID);                    // the first round takes its inputs from the loads (below)
  L9 xin[4];
  if constexpr (LAST && MID) {
    // the tile as the first pass left it: [limbs 0-3][limbs 4-7][limb 8] planes, the LDS layout itself
    const uint4* t4 = reinterpret_cast<const uint4*>(a.mid + (row << k) * 9 + (size_t)tile * (9 * T));
    uint4* l4 = reinterpret_cast<uint4*>(lds);
    if constexpr (RU >= 0) {
      // (the swizzled tile: planes 0 and 1 move whole uint4s, the limb-8 plane word by word -- (e + j) -> SWZ(e) ^ j for e = 0 mod 4)
#pragma unroll
      for (u32 i = tid; i < 2 * T; i += 256) l4[(i & ~(T - 1)) | SWZ(i & (T - 1))] = t4[i];
#pragma unroll
      for (u32 i = tid; i < T / 4; i += 256) {
        const uint4 v = t4[2 * T + i];
        const u32 b = 8 * T + SWZ(4 * i);
        lds[b] = v.x; lds[b ^ 1u] = v.y; lds[b ^ 2u] = v.z; lds[b ^ 3u] = v.w;
      }
    } else {
#pragma unroll
    for (u32 i = tid; i < 9 * T / 4; i += 256) l4[i] = t4[i];
    }
  } else {
    // The four elements a thread loads, e = tid + 256 it, ARE the quad (the two pairs) of its first round: dq = 256 there whatever the
    // pass shape.  They go to that round in registers -- no LDS round trip and, where that round needs no clamp table, no barrier
#pragma unroll
  for (u32 it = 0; it < 4; it++) {
    const u32 e = tid + 256u * it;
    const u32 g = gindex(e);
    Fe<NL> v;
    if constexpr (FIRST) {        // zero padding, the ragged tail of the caller's vector and the coeffs copy exist here only
      v = (g < a.n_valid && row * a.src_stride + g < a.n_src_total) ? fe_load<NL>(src + (size_t)g * NL) : fe_zero<NL>();
      if (a.copy_dst != nullptr && g < a.n_valid) fe_store<NL>(a.copy_dst + (row * a.src_stride + g) * NL, v);
    } else {
      v = fe_load<NL>(src + (size_t)g * NL);                 // < 2^256 (the first pass's store), not necessarily < p
    }
    xin[it] = l9::from_packed(v);
  }
  __builtin_amdgcn_sched_barrier(0);                         // (the first round's twiddle loads stay behind the conversions: registers)
  }
  const u32* cls_pack = pack + (size_t)(FIRST ? tile : 0u) * pi.class_words;
  // tiles that hold elements of "block 0" (never multiplied so far).  a.blk0_gone: an earlier pass had a uniform round and
  // converted what was left of block 0 before it (below): nothing is in Montgomery form any more
  const bool blk0_tile = (FIRST || tile == 0) && a.blk0_gone == 0;
  const bool zero_hi = FIRST && a.n_valid <= (1ull << (k - 1));
  // the limb intermediate's tile sits in LDS; otherwise only the q*p table does, which no first round reads (their pure sums are
  // sums of loads: normalised, not clamped): the barrier at the end of that round serves
  if constexpr (!DIRECT) __syncthreads();

  if constexpr (SH::U0 == 1) {
    // ---- radix-2 round at stage 0 (odd S): pairs (e1, e1 + half), twiddle w^(index of e1).  Inputs straight from the
    //      loads (< p).  Stage 0 is all block 0: with canonical output the product takes the converting table.
    constexpr u32 half = 1u << (S - 1 + LBT);
    const u32* blk = cls_pack + pi.round_off[0];
#pragma unroll
    for (u32 pp = 0; pp < 2; pp++) {
      const u32 e1 = tid + 256u * pp;                        // slots with the top stage bit clear are [0, half)
      const Fe29 w = pk_load<2>(blk, SH::period2, canon ? 1u : 0u, e1);
      const L9 x = DIRECT ? xin[pp] : lds9_get<LT>(lds, SWZ(e1));
      if (pp == 0) mem_phase(false);
      if (zero_hi) {
        if constexpr (DIRECT) lds9_put<LT>(lds, SWZ(e1), x);
        lds9_put<LT>(lds, SWZ(e1 + half), l9::mul(x, w));         // (x, 0) -> (x, x w)
      } else {
        const L9 y = DIRECT ? xin[pp + 2] : lds9_get<LT>(lds, SWZ(e1 + half));
        L9 sum = l9::add(x, y);                              // [0, 2p)
        l9::normalize(sum);
        lds9_put<LT>(lds, SWZ(e1), sum);
        lds9_put<LT>(lds, SWZ(e1 + half), l9::mul(l9::sub(x, y), w));
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
        // ---- a uniform round: wave w takes the quads q = w mod 4, whose twiddles are the three of pack slot (w): scalar operands.
        //      Block 0 is gone (converted in round RU - 1), so every lane multiplies by the same plain constants.
        const u32 wv = __builtin_amdgcn_readfirstlane(tid >> 6);
        const u32 qu = ((tid & 63u) << 2) | wv;
        const u32 lpu = qu & ((1u << LBT) - 1), ju = qu >> LBT;
        const u32 iu = ((ju >> (hb - 1)) << (hb + 1)) | (ju & ((1u << (hb - 1)) - 1));
        const u32 eu = (iu << LBT) | lpu;
        const u32* wu = cls_pack + pi.u_off + ((r - RU) * 4 + wv) * (3 * U_SLOT);
        const L9 x0 = lds9_get<LT>(lds, SWZ(eu)), x1 = lds9_get<LT>(lds, SWZ(eu + dq));
        const L9 x2 = lds9_get<LT>(lds, SWZ(eu + 2 * dq)), x3 = lds9_get<LT>(lds, SWZ(eu + 3 * dq));
        mem_phase(false);
        const L9 b0 = l9::add(x0, x2), b1 = l9::add(x1, x3);
        L9 c0 = l9::add(b0, b1);
        l9::clamp_apply(c0, l9::clamp_row(nqp, l9::clamp_q(c0.v[8])));
        lds9_put<LT>(lds, SWZ(eu), c0);
        L9 d1 = l9::sub(b0, b1);
        l9::normalize(d1);                                                                 // mul_u wants sum |limb| < 9 * 2^29
        lds9_put<LT>(lds, SWZ(eu + dq), l9::mul_u(d1, wu + 2 * U_SLOT));
        const L9 b2 = l9::mul_u(l9::sub(x0, x2), wu);
        const L9 b3 = l9::mul_u(l9::sub(x1, x3), wu + U_SLOT);                            // (-2p, 2.7p)
        L9 c2 = l9::add(b2, b3);
        l9::normalize(c2);
        lds9_put<LT>(lds, SWZ(eu + 2 * dq), c2);
        lds9_put<LT>(lds, SWZ(eu + 3 * dq), l9::mul_u(l9::sub(b2, b3), wu + 2 * U_SLOT));
        mem_phase(true);
        __syncthreads();
        continue;
      }
    }
    if (u == 0 && zero_hi) {
      // zero-padded first round (rate <= 1/2): x2 = x3 = 0, the stage-0 butterflies are (x, x w); inputs < p; everything
      // is block 0, so with canonical output the three multiplies leaving it (w0, w3, and w2 for c1) take the converting set
      const u32 vb = canon ? 3u : 0u;
      const Fe29 w0 = pk_load<6>(blk, period, vb + 0, jl), w3 = pk_load<6>(blk, period, vb + 1, jl), w2 = pk_load<6>(blk, period, vb + 2, jl);
      mem_phase(false);
      if (a.n_valid <= (1ull << (k - 2))) {                  // rate <= 1/4: x1 is zero too
        const L9 x0 = DIRECT ? xin[0] : lds9_get<LT>(lds, SWZ(e0));
        if constexpr (DIRECT) lds9_put<LT>(lds, SWZ(e0), x0);
        lds9_put<LT>(lds, SWZ(e0 + dq), l9::mul(x0, w2));
        lds9_put<LT>(lds, SWZ(e0 + 2 * dq), l9::mul(x0, w0));
        lds9_put<LT>(lds, SWZ(e0 + 3 * dq), l9::mul(x0, w3));
      } else {
        const L9 x0 = DIRECT ? xin[0] : lds9_get<LT>(lds, SWZ(e0)), x1 = DIRECT ? xin[1] : lds9_get<LT>(lds, SWZ(e0 + dq));
        L9 c0 = l9::add(x0, x1);                                                           // [0, 2p)
        l9::normalize(c0);
        lds9_put<LT>(lds, SWZ(e0), c0);
        lds9_put<LT>(lds, SWZ(e0 + dq), l9::mul(l9::sub(x0, x1), w2));
        const L9 t = l9::mul_u(x1, a.wq_w);                                                // x1 I (plain constant: t keeps x1's form); (-2p, 2.7p)
        lds9_put<LT>(lds, SWZ(e0 + 2 * dq), l9::mul(l9::add(x0, t), w0));                  // in: limbs (-2^29, 2^30), |value| < 3.7p
        lds9_put<LT>(lds, SWZ(e0 + 3 * dq), l9::mul(l9::sub(x0, t), w3));
      }
      mem_phase(true);
      __syncthreads();
      continue;
    }
    const bool from_regs = DIRECT && SH::U0 == 0 && r == 0;   // (e0 == tid, dq == 256: the thread's own loads)
    const L9 x0 = from_regs ? xin[0] : lds9_get<LT>(lds, SWZ(e0)), x1 = from_regs ? xin[1] : lds9_get<LT>(lds, SWZ(e0 + dq));
    const L9 x2 = from_regs ? xin[2] : lds9_get<LT>(lds, SWZ(e0 + 2 * dq)), x3 = from_regs ? xin[3] : lds9_get<LT>(lds, SWZ(e0 + 3 * dq));     // I: normalised, |value| < 4p
    mem_phase(false);
    const L9 b0 = l9::add(x0, x2), b1 = l9::add(x1, x3);                                   // limbs [0, 2^30), |value| < 8p
    L9 c0 = l9::add(b0, b1);                                                               // limbs [0, 2^31), |value| < 16p
    if (last_two) {
      // outputs go straight to the store path (normalised, |value| < 16p)
      l9::normalize(c0);
      L9 c1 = l9::sub(b0, b1);
      const L9 b2 = l9::sub(x0, x2);
      // w^(n/4) is the one twiddle every lane shares: its multiply takes the shifted-multiples form (scalar operands)
      const L9 b3 = l9::mul_u(l9::sub(x1, x3), a.wq_w);
      L9 c2 = l9::add(b2, b3);
      L9 c3 = l9::sub(b2, b3);
      l9::normalize(c1); l9::normalize(c2); l9::normalize(c3);
      // dq == 1 here: the quad is four CONSECUTIVE elements of the row, 128 bytes -- reduced and stored from the registers (no LDS round
      // trip, no barrier; -0.2 ... -0.8 % by shape).  -> [0, p): after the clamp, value >= p needs the top limb to reach floor(p / 2^232)
      // -- about one element in 2^17; the conditional subtract runs only in the waves that hold such an element
      mem_phase(true);
      u32* dstq = a.dst + row * a.dst_stride * NL + (size_t)((tile << S) | e0) * NL;
      L9 cc[4] = {c0, c1, c2, c3};
#pragma unroll
      for (int c = 0; c < 4; c++) {
        L9 x = cc[c];                                                     // normalised, |value| < 16p
        l9::clamp_apply(x, l9::clamp_row(nqp, l9::clamp_q(x.v[8])));      // [0, p + 2^239) < 2^256
        u32 w[8];
        fe_from29(w, x.v);
        Fe<NL> v;
#pragma unroll
        for (int i = 0; i < 8; i++) v.v[i] = w[i];
        if (__any((int)(x.v[8] >= (u32)P29::limb(8)))) v = fe_reduce_once8(w);
        fe_store<NL>(dstq + (size_t)c * NL, v);
      }
      return;
    } else {
      // clamp the pure sum at once: c0 leaves the registers before the multiplier chains start (holding it and its
      // q*p row across them spills at 128 VGPRs: +1 GB of scratch writes per pass, profiles/r02b)
      // (a first round fed by the loads: four values < p + 2^239, their sum < 4.001 p needs the carries only -- and no q*p table yet)
      if (from_regs) l9::normalize(c0);
      else l9::clamp_apply(c0, l9::clamp_row(nqp, l9::clamp_q(c0.v[8])));                  // [0, p + 2^239)
      lds9_put<LT>(lds, SWZ(e0), c0);
      // block 0 of stages (u, u + 1) = the quads whose elements all lie below n / 2^(t + 2): here exactly q < period in the
      // tiles that hold block 0.  Their three multiplies that leave block 0 (c1, c2, c3) take the converting set; c0 stays a
      // pure sum
      const bool blk0c = canon && blk0_tile && q < period && (RU < 0 || r < RU);
      if constexpr (RU >= 1) {
        if (r == RU - 1 && blk0c) {
          // the last round before the uniform one: c0, the pure sum that would carry block 0 on, is converted as well (a multiply
          // by 2^5 = 2^261 R^-1: 16 lanes of one wave per block-0 tile), so that the uniform round sees canonical values only
          Fe29 k32;
#pragma unroll
          for (int i = 0; i < 9; i++) k32.v[i] = i == 0 ? 32u : 0u;
          lds9_put<LT>(lds, SWZ(e0), l9::mul(c0, k32));
        }
      }
      const u32 vb = blk0c ? 3u : 0u;
      const Fe29 w0 = pk_load<6>(blk, period, vb + 0, jl), w3 = pk_load<6>(blk, period, vb + 1, jl);
      const Fe29 w2 = pk_load<6>(blk, period, vb + 2, jl);
      const L9 d1 = l9::sub(b0, b1);                                                       // limbs (-2^30, 2^30), |value| < 16p
      lds9_put<LT>(lds, SWZ(e0 + dq), l9::mul(d1, w2));                                    // normalised, (-1.2p, 0.2p]
      // t = (x1 - x3) I by the plain constant: in block 0 it stays in the form of its inputs and the two products below convert
      const L9 t = l9::mul_u(l9::sub(x1, x3), a.wq_w);                                     // normalised, (-2p, 2.7p)
      const L9 e2 = l9::sub(x0, x2);                                                       // limbs (-2^29, 2^29), |value| < 8p
      lds9_put<LT>(lds, SWZ(e0 + 2 * dq), l9::mul(l9::add(e2, t), w0));                    // in: limbs (-2^29, 2^30), |value| < 10.7p
      lds9_put<LT>(lds, SWZ(e0 + 3 * dq), l9::mul(l9::sub(e2, t), w3));
    }
    mem_phase(true);                                         // the barrier, then the next round's reads and twiddle loads (or the store phase)
    __syncthreads();
  }

  if constexpr (FIRST && MID) {
    // every slot holds a normalised value with |value| < 4p (c0 clamped, c1 / c3 products, c2 a normalised sum of two
    // products; after a lone radix-2 round: loads, normalised sums, products): stored as it is.  lb == 10 (the successor
    // runs 10 stages), so slot (i, lp) is element (tile << LTJ | lp) of the successor's tile i
    u32* mrow = a.mid + (row << k) * 9;
#pragma unroll
    for (u32 e = tid; e < T; e += 256) {
      const u32 e2 = (tile << LTJ) | (e & ((1u << LBT) - 1));
      u32* t = mrow + (size_t)(e >> LBT) * (9 * T);
      const L9 x = lds9_get<LT>(lds, SWZ(e));
      *reinterpret_cast<uint4*>(t + (size_t)e2 * 4) = make_uint4(x.v[0], x.v[1], x.v[2], x.v[3]);
      *reinterpret_cast<uint4*>(t + (size_t)(T + e2) * 4) = make_uint4(x.v[4], x.v[5], x.v[6], x.v[7]);
      t[(size_t)8 * T + e2] = x.v[8];
    }
    return;
  }
  if constexpr (FIRST) {           // (a last pass has stored from its final round and returned)
  u32* dst = a.dst + row * a.dst_stride * NL;
#pragma unroll
  for (u32 e = tid; e < T; e += 256) {
    const u32 g = gindex(e);
    L9 x = lds9_get<LT>(lds, SWZ(e));                                        // normalised, |value| < 16p
    l9::clamp_apply(x, l9::clamp_row(nqp, l9::clamp_q(x.v[8])));        // [0, p + 2^239) < 2^256: what the successor reads as limbs anyway
    u32 w[8];
    fe_from29(w, x.v);
    Fe<NL> v;
#pragma unroll
    for (int i = 0; i < 8; i++) v.v[i] = w[i];
    fe_store<NL>(dst + (size_t)g * NL, v);
  }
  }
}

// one thread per (class, round slot, position): copies the table entries a quad / pair will ask for into lane order
template <int S, int LBT>
__global__ void __launch_bounds__(256) ntt_pack_kernel(NttPassArgs a, NttPackInfo pi, u32 n_classes, bool first, u32* pack) {
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
      const u32 idx = (g1 & gm) << t0;
      for (u32 v = 0; v < 2; v++) {
        const u32* e = (v == 0 ? a.roots29 : a.roots29c) + (size_t)idx * 12;
        for (u32 c = 0; c < 2; c++)
          for (u32 w = 0; w < 4; w++) blk[((size_t)(v * 2 + c) * SH::period2 + jl) * 4 + w] = e[c * 4 + w];
        blk[(size_t)2 * 2 * SH::period2 * 4 + (size_t)v * SH::period2 + jl] = e[8];
      }
      continue;
    }
    const u32 r = slot - SH::U0, u = SH::U0 + 2 * r, hb = S - u - 1, period = 1u << (hb - 1 + LBT);
    const u32 t = t0 + u;
    if (jl >= period || t + 2 == k) continue;                // (stages k-2, k-1: one wave-uniform twiddle, not packed)
    const u32 gm0 = (1u << (k - t - 1)) - 1;
    const u32 lp = jl & ((1u << LBT) - 1), j = jl >> LBT;
    const u32 i0 = ((j >> (hb - 1)) << (hb + 1)) | (j & ((1u << (hb - 1)) - 1));
    const u32 g0 = (i0 << lb) | lo | lp;                     // (last pass: the tile's high bits do not reach these twiddles)
    // w0 = w^e, w3 = w^(3 e), w2 = w^(2 e) with e = (g0 & gm0) << t < n / 4 (g0 has the quarter bit clear; the pass kernel's
    // w1 = w^(e + n/4) = I w0 is never read).  The tables hold w^i for i < n / 2 and w^(n/2) = -1: past that, the negated entry
    const u32 e1x = (g0 & gm0) << t, half_n = 1u << (k - 1);
    const u32 idx[3] = {e1x, 3 * e1x, 2 * e1x};
    for (u32 v = 0; v < 6; v++) {
      const u32 ix = idx[v % 3];
      const u32* e = (v < 3 ? a.roots29 : a.roots29c) + (size_t)(ix & (half_n - 1)) * 12;
      u32 m[9];
      if (ix >= half_n) {                                    // p - entry, limb-wise with borrow (entry in (0, p))
        int32_t br = 0;
#pragma unroll
        for (int z = 0; z < 9; z++) {
          const int32_t d = (int32_t)P29::limb(z) - (int32_t)e[z] - br;
          br = d < 0 ? 1 : 0;
          m[z] = z < 8 ? (u32)d & P29::M : (u32)d;
        }
      } else {
#pragma unroll
        for (int z = 0; z < 9; z++) m[z] = e[z];
      }
      for (u32 c = 0; c < 2; c++)
        for (u32 w = 0; w < 4; w++) blk[((size_t)(v * 2 + c) * period + jl) * 4 + w] = m[c * 4 + w];
      blk[(size_t)6 * 2 * period * 4 + (size_t)v * period + jl] = m[8];
    }
  }
}

// the uniform round's constants: per class, for jl = 0..3 and the round's three twiddles w0, w1, w2 (plain: block 0 is gone by then),
// the nine shifted multiples W_j = balanced(w 2^(29 j) mod p) as 81 words t = 9 k + j (limb k of W_j; l9::mul_u / field_wmul_gen.h).
// The table entry is w 2^261 mod p as limbs: fe_mul_r29(2^(29 j), entry) = w 2^(29 j), fully reduced.
template <int S, int LBT>
__global__ void __launch_bounds__(64) ntt_upack_kernel(NttPassArgs a, NttPackInfo pi, u32 n_classes, bool first, u32* pack) {
  using SH = Shape<S, LBT>;
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
  const Fe29 w = tw_entry29(a.roots29, idx);
  u32* out = pack + (size_t)cls * pi.class_words + pi.u_off + ((ru * 4 + jl0) * 3 + v) * U_SLOT;
  for (u32 jj = 0; jj < 9; jj++) {
    Fe<8> sh = fe_zero<8>();
    sh.v[(29 * jj) / 32] = 1u << ((29 * jj) % 32);
    const Fe<8> val = fe_mul_r29(sh, w);                     // w 2^(29 jj) mod p, in [0, p)
    u32 m[9];
#pragma unroll
    for (int z = 0; z < 8; z++) m[z] = val.v[z];
    m[8] = 0;
    bool big = false, decided = false;                       // val > (p - 1) / 2 -> val - p (288-bit two's complement)
#pragma unroll
    for (int z = 7; z >= 0; z--) {
      const u32 hz = (Mod<8>::P[z] >> 1) | (z < 7 ? (Mod<8>::P[z + 1] & 1u) << 31 : 0u);
      if (!decided && val.v[z] != hz) { big = val.v[z] > hz; decided = true; }
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
      out[9 * kk + jj] = kk < 8 ? (u32)(x & P29::M) : (u32)x;  // limb 8: bits 232 .. 263, sign-extended (m[8] is 0 or ~0)
    }
  }
  for (u32 z = 81; z < U_SLOT; z++) out[z] = 0;
}

template <int S, int LBT> NttPackInfo pack_info_t() {
  using SH = Shape<S, LBT>;
  NttPackInfo pi{};
  u32 off = 0, slot = 0;
  if (SH::U0) { pi.round_off[slot++] = off; off += 2 * SH::period2 * 9; off = (off + 3) & ~3u; }
  for (int r = 0; r < SH::NR4; r++) { pi.round_off[slot++] = off; off += 6 * SH::period4(r) * 9; off = (off + 3) & ~3u; }
  if (SH::RU >= 0) { pi.u_off = off; off += SH::NRU * 4 * 3 * U_SLOT; }
  pi.class_words = off;
  return pi;
}

template <int S, int LTJ, bool FIRST, bool MID>
hipError_t launch_tm(const NttPassArgs& a, const u32* pack, const NttPackInfo& pi, hipStream_t st) {
  constexpr int LT = S + LTJ;
  const u64 tiles = ((u64)1 << (a.log_n - LT)) * a.n_rows;
  const size_t lds_bytes = (size_t)Lds9<LT>::WORDS * 4;
  // hipFuncSetAttribute is idempotent and cheap; calling it on every launch keeps this free of unsynchronised caches
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&ntt_pass_l9s_kernel<S, LTJ, FIRST, MID>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL((ntt_pass_l9s_kernel<S, LTJ, FIRST, MID>), dim3((unsigned)tiles), dim3(256), lds_bytes, st, a, pack, pi);
  return hipGetLastError();
}
template <int S, int LTJ, bool FIRST>
hipError_t launch_t(const NttPassArgs& a, const u32* pack, const NttPackInfo& pi, hipStream_t st) {
  return a.mid ? launch_tm<S, LTJ, FIRST, true>(a, pack, pi, st) : launch_tm<S, LTJ, FIRST, false>(a, pack, pi, st);
}

}  // namespace

bool ntt_l9s3_supported(uint32_t log_n) { return log_n >= 21 && log_n <= 26; }

__global__ void __launch_bounds__(256) subtable_kernel(const u32* tab, u32 shift, u64 n, u32* sub) {
  for (u64 id = (u64)blockIdx.x * 256 + threadIdx.x; id < n * 3; id += (u64)gridDim.x * 256) {
    const u64 i = id / 3, c = id % 3;
    reinterpret_cast<uint4*>(sub)[i * 3 + c] = reinterpret_cast<const uint4*>(tab)[(i << shift) * 3 + c];
  }
}
hipError_t launch_ntt_l9s_subtable(const uint32_t* tab, uint32_t shift, uint64_t n, uint32_t* sub, hipStream_t st) {
  hipLaunchKernelGGL(subtable_kernel, dim3(2048), dim3(256), 0, st, tab, shift, n, sub);
  return hipGetLastError();
}

bool ntt_l9s_supported(uint32_t log_n, uint32_t n_passes, int log_tile) { return n_passes == 2 && log_tile == 10 && log_n >= 11 && log_n <= 20; }

#define L9S_FIRST_CASES(X) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10)

NttPackInfo ntt_l9s_pack_info(uint32_t s, bool first) {
  if (!first) return pack_info_t<10, 0>();
  switch (s) {
#define X(SV) case SV: return pack_info_t<SV, 10 - SV>();
    L9S_FIRST_CASES(X)
#undef X
  }
  return NttPackInfo{};
}

hipError_t launch_ntt_l9s_pack(const NttPassArgs& a, bool first, const NttPackInfo& pi, uint32_t n_classes, uint32_t* pack, hipStream_t st) {
  const unsigned grid = 2048;
  const unsigned ugrid = (n_classes * 2 * 12 + 63) / 64;      // (<= 2 uniform rounds per pass)
  if (!first) {
    hipLaunchKernelGGL((ntt_pack_kernel<10, 0>), dim3(64), dim3(256), 0, st, a, pi, n_classes, false, pack);
    hipLaunchKernelGGL((ntt_upack_kernel<10, 0>), dim3(ugrid), dim3(64), 0, st, a, pi, n_classes, false, pack);
    return hipGetLastError();
  }
  switch (a.s) {
#define X(SV) case SV: hipLaunchKernelGGL((ntt_pack_kernel<SV, 10 - SV>), dim3(grid), dim3(256), 0, st, a, pi, n_classes, true, pack); break;
    L9S_FIRST_CASES(X)
#undef X
    default: return hipErrorInvalidValue;
  }
  if (a.s == 8) hipLaunchKernelGGL((ntt_upack_kernel<8, 2>), dim3(ugrid), dim3(64), 0, st, a, pi, n_classes, true, pack);
  if (a.s == 9) hipLaunchKernelGGL((ntt_upack_kernel<9, 1>), dim3(ugrid), dim3(64), 0, st, a, pi, n_classes, true, pack);
  if (a.s == 10) hipLaunchKernelGGL((ntt_upack_kernel<10, 0>), dim3(ugrid), dim3(64), 0, st, a, pi, n_classes, true, pack);
  return hipGetLastError();
}

hipError_t launch_ntt_pass_l9s(const NttPassArgs& a, bool first, const uint32_t* pack, const NttPackInfo& pi, hipStream_t st) {
  if (!first) {
    if (a.s != 10 || a.log_tj != 0 || a.t0 + a.s != a.log_n) return hipErrorInvalidValue;
    // (the last pass has a uniform round and converts block 0 before it: no Montgomery-form prefix is left for the store to reduce)
    if (a.mont_prefix) return hipErrorInvalidValue;
    return launch_t<10, 0, false>(a, pack, pi, st);
  }
  // two-pass plans: s + 10 stages in all; three-pass plans: s + 20 (the first pass works at element stride 2^20)
  if (a.t0 != 0 || a.s + a.log_tj != 10 || (a.s + 10 != a.log_n && a.s + 20 != a.log_n)) return hipErrorInvalidValue;
  if (a.tile_group && ((1u << (a.log_n - 10)) >> a.tile_group) < 8) return hipErrorInvalidValue;
  switch (a.s) {
#define X(SV) case SV: return launch_t<SV, 10 - SV, true>(a, pack, pi, st);
    L9S_FIRST_CASES(X)
#undef X
  }
  return hipErrorInvalidValue;
}

}  // namespace lcpc
