// lcpc_amd/csrc/host_crypto.h -- host-side Fiat-Shamir primitives used by prove/verify:
// merlin::Transcript (STROBE-128 / Keccak-f[1600]), ChaCha20Rng, rand's Uniform, ff's Field::random,
// and a host BLAKE3 for the verifier's Merkle-path check.  These are inherently serial
// (lcpc-2d/src/lib.rs:1022-1080) and stay on the host, as SURVEY.md 3.3 notes.
#pragma once
#include <stddef.h>
#include <stdint.h>
#include <vector>
#include "host_field.h"

namespace lcpc {

void keccak_f1600(uint64_t a[25]);            // AVX-512 when the CPU has it (run-time check), else the portable form
void keccak_f1600_portable(uint64_t a[25]);   // the scalar form, always (tests compare the two)

// merlin 2.0 Transcript [3P]
class Transcript {
 public:
  explicit Transcript(const uint8_t* label, size_t len);
  void append_message(const uint8_t* label, size_t llen, const uint8_t* msg, size_t mlen);
  // n calls of append_message(label, msgs + i * mlen, mlen), i = 0..n-1: the shape of prove / verify's coefficient
  // absorbs (lcpc-2d lib.rs:1045-1047, 1066-1068)
  void append_messages(const uint8_t* label, size_t llen, const uint8_t* msgs, size_t mlen, size_t n);
  void challenge_bytes(const uint8_t* label, size_t llen, uint8_t* out, size_t n);

 private:
  static constexpr int R = 166;
  union { uint8_t b[200]; uint64_t w[25]; } st_;
  uint8_t pos_ = 0, pos_begin_ = 0, cur_flags_ = 0;
  void run_f();
  void absorb(const uint8_t* d, size_t n);
  void squeeze(uint8_t* d, size_t n);
  void begin_op(uint8_t flags, bool more);
  void meta_ad(const uint8_t* d, size_t n, bool more);
  void ad(const uint8_t* d, size_t n, bool more);
  void prf(uint8_t* d, size_t n, bool more);
};

// ChaCha20 keystream (the rand_chacha layout: 64-bit block counter in words 12-13, 64-bit stream id in words 14-15):
// the 16 * nblocks words of blocks [block0, block0 + nblocks).  The stream is seekable, so long runs are generated in
// bulk, 16 blocks per vector pass and on as many threads as the caller likes (matgen).
void chacha20_keystream(const uint32_t key[8], uint64_t stream, uint64_t block0, uint64_t nblocks, uint32_t* out);

// rand_chacha 0.3 ChaCha20Rng [3P]: 64-bit block counter, 64-bit stream id, 4-block buffer
class ChaCha20Rng {
 public:
  explicit ChaCha20Rng(const uint8_t seed[32]);
  static ChaCha20Rng seed_from_u64(uint64_t state);     // rand_core 0.6 PCG32 expansion [3P]
  void set_stream(uint64_t s) { stream_ = s; }
  const uint32_t* key() const { return key_; }
  uint64_t stream() const { return stream_; }
  uint32_t next_u32();
  uint64_t next_u64();
  uint64_t uniform(uint64_t high);                      // rand 0.8 Uniform::<usize>::new(0, high).sample [3P]
  void field_random(const FieldDesc& f, uint64_t* out); // ff_derive Field::random [3P]

 private:
  uint32_t key_[8];
  uint64_t counter_ = 0, stream_ = 0;
  uint32_t buf_[64];
  int idx_ = 64;
  void refill();
};

// BLAKE3 (plain hash, 32-byte output) of a contiguous message
void blake3_host(const uint8_t* in, size_t len, uint8_t out[32]);

}  // namespace lcpc
