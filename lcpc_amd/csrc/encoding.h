// lcpc_amd/csrc/encoding.h -- host side of the two LcEncoding implementors: parameter selection
// (matrix shape, #column openings, #degree tests) and, for Brakedown, deterministic generation of the
// expander matrices.  Mirrors /root/reference/lcpc-ligero-pc/src/lib.rs:45-148 and
// lcpc-brakedown-pc/src/{lib.rs:54-137, matgen.rs, codespec.rs}.
#pragma once
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <new>
#include <vector>
#include "host_field.h"

namespace lcpc {

uint64_t log2_ceil(uint64_t v);                                        // lcpc-2d lib.rs:827-829
uint64_t next_pow2(uint64_t v);
uint64_t n_degree_tests(uint64_t lambda, uint64_t len, uint64_t flog2);  // lcpc-2d lib.rs:613-616

// Ligero
uint64_t ligero_n_col_opens(uint32_t rho_num, uint32_t rho_den);        // ligero lib.rs:61-64
// _get_dims (ligero lib.rs:70-112): 0 on success, <0 if n_cols would exceed 2^S
int ligero_get_dims(const FieldDesc& f, uint64_t len, uint32_t rho_num, uint32_t rho_den, uint64_t* n_rows,
                    uint64_t* n_per_row, uint64_t* n_cols);

// Brakedown / SDIG
struct SdigSpec {
  uint64_t an, ad, bn, bd, rn, rd, baselen;
  double alpha, beta, r, dist, mu, nu, cn1, cn2, dn1, dn2;
};
bool sdig_spec(int code, SdigSpec* s);                                   // codespec.rs:24-129, 169-232
uint64_t sdig_n_col_opens(int code);                                     // brakedown lib.rs:57-61
struct LevelDims { uint64_t n, m, d; };                                  // (inputs, outputs, nnz per input)
// matgen.rs:56-111; false if n <= baselen
bool sdig_level_dims(const SdigSpec& s, uint64_t n, double log2p, std::vector<LevelDims>& pre, std::vector<LevelDims>& post);
uint64_t sdig_codeword_length(const std::vector<LevelDims>& pre, const std::vector<LevelDims>& post);  // encode.rs:18-33
// SdigEncoding::new's choice of n_per_row (brakedown lib.rs:103-110 -> 69-87)
bool sdig_n_per_row(const FieldDesc& f, uint64_t len, int code, uint64_t* n_per_row, bool ml = false);

// plain array without value-initialisation: the matrices are tens of megabytes that are overwritten anyway, and a
// std::vector would first zero them on ONE thread (page faults included) before the parallel fill starts
template <typename T> struct RawBuf {
  T* p = nullptr;
  size_t n = 0;
  RawBuf() = default;
  RawBuf(const RawBuf&) = delete;
  RawBuf& operator=(const RawBuf&) = delete;
  RawBuf(RawBuf&& o) noexcept : p(o.p), n(o.n) { o.p = nullptr; o.n = 0; }
  RawBuf& operator=(RawBuf&& o) noexcept { if (this != &o) { free(p); p = o.p; n = o.n; o.p = nullptr; o.n = 0; } return *this; }
  ~RawBuf() { free(p); }
  void alloc(size_t count) {
    free(p);
    p = static_cast<T*>(malloc((count ? count : 1) * sizeof(T)));
    if (!p) throw std::bad_alloc();
    n = count;
  }
  void zero() { memset(p, 0, n * sizeof(T)); }
  size_t size() const { return n; }
  T* data() { return p; }
  const T* data() const { return p; }
  T& operator[](size_t i) { return p[i]; }
  const T& operator[](size_t i) const { return p[i]; }
};
// one expander matrix, CSR by *output* (the transpose view of the reference's CSC-by-input)
struct CsrMatrix {
  uint64_t n_in = 0, n_out = 0;
  RawBuf<uint32_t> rowptr;    // n_out + 1
  RawBuf<uint32_t> colidx;    // nnz, input index
  RawBuf<uint64_t> vals;      // nnz * L, Montgomery limbs
};
// matgen.rs:28-52 + 114-188: precode[i], postcode[i] from (n_per_row, seed); RNG-order identical to the
// reference (per level: ChaCha20Rng::seed_from_u64(seed), set_stream(i); precode then postcode).
bool sdig_generate(const FieldDesc& f, const SdigSpec& s, uint64_t n_per_row, uint64_t seed, std::vector<CsrMatrix>& pre,
                   std::vector<CsrMatrix>& post, std::vector<LevelDims>& pre_dims, std::vector<LevelDims>& post_dims);

}  // namespace lcpc
