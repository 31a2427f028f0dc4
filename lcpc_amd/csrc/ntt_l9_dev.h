// lcpc_amd/csrc/ntt_l9_dev.h -- LDS tile access shared by the two Ft255 lazy-limb NTT kernels
// (ntt_pass_l9_kernel in kernels.hip, ntt_pass_l9s_kernel in ntt_l9s.hip).
//
// A tile of 2^LT elements lives in LDS in the multiplier's own format (field_dev.h, namespace l9): 9 limbs of 29 bits,
// chunk-major -- limbs 0..3 of element e at word 4e, limbs 4..7 at 4(T + e), limb 8 at 8T + e -- so that an element is
// two ds_read_b128 and one ds_read_b32.  The q*p table of l9::clamp follows the tile.
#pragma once
#include "field_dev.h"

namespace lcpc {

template <int LT> struct Lds9 {
  static constexpr u32 T = 1u << LT;
  static constexpr u32 WORDS = T * 9 + 64 * 12;              // tile + q*p table (64 entries, 12-word stride)
};
template <int LT> LCPC_DEV L9 lds9_get(const u32* lds, u32 e) {
  const uint4 a = *reinterpret_cast<const uint4*>(lds + (size_t)e * 4);
  const uint4 b = *reinterpret_cast<const uint4*>(lds + ((size_t)Lds9<LT>::T + e) * 4);
  L9 r;
  r.v[0] = a.x; r.v[1] = a.y; r.v[2] = a.z; r.v[3] = a.w; r.v[4] = b.x; r.v[5] = b.y; r.v[6] = b.z; r.v[7] = b.w;
  r.v[8] = lds[(size_t)Lds9<LT>::T * 8 + e];
  return r;
}
template <int LT> LCPC_DEV void lds9_put(u32* lds, u32 e, const L9& x) {
  *reinterpret_cast<uint4*>(lds + (size_t)e * 4) = make_uint4(x.v[0], x.v[1], x.v[2], x.v[3]);
  *reinterpret_cast<uint4*>(lds + ((size_t)Lds9<LT>::T + e) * 4) = make_uint4(x.v[4], x.v[5], x.v[6], x.v[7]);
  lds[(size_t)Lds9<LT>::T * 8 + e] = x.v[8];
}
// one entry of a 12-word-stride twiddle table (roots29 / roots29c)
LCPC_DEV Fe29 tw_entry29(const u32* tab, u32 widx) {
  Fe29 t;
  const uint4* wp = reinterpret_cast<const uint4*>(tab + (size_t)widx * 12);
  const uint4 w0 = wp[0], w1 = wp[1];
  const u32 w8 = tab[(size_t)widx * 12 + 8];
  t.v[0] = w0.x; t.v[1] = w0.y; t.v[2] = w0.z; t.v[3] = w0.w;
  t.v[4] = w1.x; t.v[5] = w1.y; t.v[6] = w1.z; t.v[7] = w1.w; t.v[8] = w8;
  return t;
}

}  // namespace lcpc
