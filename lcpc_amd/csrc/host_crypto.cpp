// lcpc_amd/csrc/host_crypto.cpp -- see host_crypto.h
#include "host_crypto.h"
#include <string.h>

namespace lcpc {

static inline uint64_t rol64(uint64_t x, int n) { return n ? (x << n) | (x >> (64 - n)) : x; }
static inline uint32_t ror32(uint32_t x, int n) { return (x >> n) | (x << (32 - n)); }

void keccak_f1600(uint64_t a[25]) {
  static const uint64_t RC[24] = {
      0x0000000000000001ull, 0x0000000000008082ull, 0x800000000000808Aull, 0x8000000080008000ull, 0x000000000000808Bull,
      0x0000000080000001ull, 0x8000000080008081ull, 0x8000000000008009ull, 0x000000000000008Aull, 0x0000000000000088ull,
      0x0000000080008009ull, 0x000000008000000Aull, 0x000000008000808Bull, 0x800000000000008Bull, 0x8000000000008089ull,
      0x8000000000008003ull, 0x8000000000008002ull, 0x8000000000000080ull, 0x000000000000800Aull, 0x800000008000000Aull,
      0x8000000080008081ull, 0x8000000000008080ull, 0x0000000080000001ull, 0x8000000080008008ull};
  // rho offsets indexed by x + 5y, pi destination computed on the fly
  static const int RHO[25] = {0, 1, 62, 28, 27, 36, 44, 6, 55, 20, 3, 10, 43, 25, 39, 41, 45, 15, 21, 8, 18, 2, 61, 56, 14};
  for (int round = 0; round < 24; round++) {
    uint64_t c[5], b[25];
    for (int x = 0; x < 5; x++) c[x] = a[x] ^ a[x + 5] ^ a[x + 10] ^ a[x + 15] ^ a[x + 20];
    for (int x = 0; x < 5; x++) {
      const uint64_t d = c[(x + 4) % 5] ^ rol64(c[(x + 1) % 5], 1);
      for (int y = 0; y < 25; y += 5) a[x + y] ^= d;
    }
    for (int x = 0; x < 5; x++)
      for (int y = 0; y < 5; y++) b[y + 5 * ((2 * x + 3 * y) % 5)] = rol64(a[x + 5 * y], RHO[x + 5 * y]);
    for (int y = 0; y < 25; y += 5)
      for (int x = 0; x < 5; x++) a[x + y] = b[x + y] ^ (~b[(x + 1) % 5 + y] & b[(x + 2) % 5 + y]);
    a[0] ^= RC[round];
  }
}

enum { SF_I = 1, SF_A = 2, SF_C = 4, SF_T = 8, SF_M = 16, SF_K = 32 };

Transcript::Transcript(const uint8_t* label, size_t len) {
  memset(st_.b, 0, 200);
  const uint8_t hdr[6] = {1, R + 2, 1, 0, 1, 96};
  memcpy(st_.b, hdr, 6);
  memcpy(st_.b + 6, "STROBEv1.0.2", 12);
  keccak_f1600(st_.w);
  meta_ad(reinterpret_cast<const uint8_t*>("Merlin v1.0"), 11, false);
  append_message(reinterpret_cast<const uint8_t*>("dom-sep"), 7, label, len);
}
void Transcript::run_f() {
  st_.b[pos_] ^= pos_begin_;
  st_.b[pos_ + 1] ^= 0x04;
  st_.b[R + 1] ^= 0x80;
  keccak_f1600(st_.w);
  pos_ = 0;
  pos_begin_ = 0;
}
void Transcript::absorb(const uint8_t* d, size_t n) {
  for (size_t i = 0; i < n; i++) {
    st_.b[pos_++] ^= d[i];
    if (pos_ == R) run_f();
  }
}
void Transcript::squeeze(uint8_t* d, size_t n) {
  for (size_t i = 0; i < n; i++) {
    d[i] = st_.b[pos_];
    st_.b[pos_++] = 0;
    if (pos_ == R) run_f();
  }
}
void Transcript::begin_op(uint8_t flags, bool more) {
  if (more) return;
  const uint8_t hdr[2] = {pos_begin_, flags};
  pos_begin_ = pos_ + 1;
  cur_flags_ = flags;
  absorb(hdr, 2);
  if ((flags & (SF_C | SF_K)) && pos_ != 0) run_f();
}
void Transcript::meta_ad(const uint8_t* d, size_t n, bool more) { begin_op(SF_M | SF_A, more); absorb(d, n); }
void Transcript::ad(const uint8_t* d, size_t n, bool more) { begin_op(SF_A, more); absorb(d, n); }
void Transcript::prf(uint8_t* d, size_t n, bool more) { begin_op(SF_I | SF_A | SF_C, more); squeeze(d, n); }
void Transcript::append_message(const uint8_t* label, size_t llen, const uint8_t* msg, size_t mlen) {
  const uint32_t l = (uint32_t)mlen;
  const uint8_t le[4] = {(uint8_t)l, (uint8_t)(l >> 8), (uint8_t)(l >> 16), (uint8_t)(l >> 24)};
  meta_ad(label, llen, false);
  meta_ad(le, 4, true);
  ad(msg, mlen, false);
}
void Transcript::challenge_bytes(const uint8_t* label, size_t llen, uint8_t* out, size_t n) {
  const uint32_t l = (uint32_t)n;
  const uint8_t le[4] = {(uint8_t)l, (uint8_t)(l >> 8), (uint8_t)(l >> 16), (uint8_t)(l >> 24)};
  meta_ad(label, llen, false);
  meta_ad(le, 4, true);
  prf(out, n, false);
}

// ---- ChaCha20 ------------------------------------------------------------------------------------
ChaCha20Rng::ChaCha20Rng(const uint8_t seed[32]) { memcpy(key_, seed, 32); }
ChaCha20Rng ChaCha20Rng::seed_from_u64(uint64_t state) {
  uint8_t seed[32];
  for (int i = 0; i < 8; i++) {
    state = state * 6364136223846793005ull + 11634580027462260723ull;
    const uint32_t xs = (uint32_t)(((state >> 18) ^ state) >> 27);
    const uint32_t rot = (uint32_t)(state >> 59);
    const uint32_t x = rot ? ror32(xs, rot) : xs;
    memcpy(seed + 4 * i, &x, 4);
  }
  return ChaCha20Rng(seed);
}
void ChaCha20Rng::refill() {
  for (int blk = 0; blk < 4; blk++) {
    uint32_t s[16] = {0x61707865u, 0x3320646Eu, 0x79622D32u, 0x6B206574u};
    for (int i = 0; i < 8; i++) s[4 + i] = key_[i];
    s[12] = (uint32_t)counter_; s[13] = (uint32_t)(counter_ >> 32);
    s[14] = (uint32_t)stream_;  s[15] = (uint32_t)(stream_ >> 32);
    uint32_t x[16];
    memcpy(x, s, 64);
    auto qr = [&](int a, int b, int c, int d) {
      x[a] += x[b]; x[d] = ror32(x[d] ^ x[a], 16);
      x[c] += x[d]; x[b] = ror32(x[b] ^ x[c], 20);
      x[a] += x[b]; x[d] = ror32(x[d] ^ x[a], 24);
      x[c] += x[d]; x[b] = ror32(x[b] ^ x[c], 25);
    };
    for (int r = 0; r < 10; r++) {
      qr(0, 4, 8, 12); qr(1, 5, 9, 13); qr(2, 6, 10, 14); qr(3, 7, 11, 15);
      qr(0, 5, 10, 15); qr(1, 6, 11, 12); qr(2, 7, 8, 13); qr(3, 4, 9, 14);
    }
    for (int i = 0; i < 16; i++) buf_[16 * blk + i] = x[i] + s[i];
    counter_++;
  }
  idx_ = 0;
}
uint32_t ChaCha20Rng::next_u32() {
  if (idx_ >= 64) refill();
  return buf_[idx_++];
}
uint64_t ChaCha20Rng::next_u64() {   // rand_core BlockRng::next_u64
  uint64_t lo, hi;
  if (idx_ < 63) { lo = buf_[idx_]; hi = buf_[idx_ + 1]; idx_ += 2; }
  else if (idx_ >= 64) { refill(); lo = buf_[0]; hi = buf_[1]; idx_ = 2; }
  else { lo = buf_[63]; refill(); hi = buf_[0]; idx_ = 1; }
  return lo | (hi << 32);
}
uint64_t ChaCha20Rng::uniform(uint64_t range) {
  const uint64_t ints_to_reject = (UINT64_MAX - range + 1) % range;
  const uint64_t zone = UINT64_MAX - ints_to_reject;
  for (;;) {
    const uint64_t v = next_u64();
    const u128 m = (u128)v * range;
    if ((uint64_t)m <= zone) return (uint64_t)(m >> 64);
  }
}
void ChaCha20Rng::field_random(const FieldDesc& f, uint64_t* out) {
  for (;;) {
    for (int i = 0; i < f.L; i++) out[i] = next_u64();
    out[f.L - 1] &= f.top_mask;
    if (!h_ge_p(f, out)) return;       // accepted raw limbs are the Montgomery representation
  }
}

// ---- BLAKE3 (host) ---------------------------------------------------------------------------------
static const uint32_t IV[8] = {0x6A09E667u, 0xBB67AE85u, 0x3C6EF372u, 0xA54FF53Au, 0x510E527Fu, 0x9B05688Cu, 0x1F83D9ABu, 0x5BE0CD19u};
static void compress(uint32_t cv[8], const uint32_t blk[16], uint64_t counter, uint32_t blen, uint32_t flags) {
  static const uint8_t SCHED[7][16] = {
      {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15}, {2, 6, 3, 10, 7, 0, 4, 13, 1, 11, 12, 5, 9, 14, 15, 8},
      {3, 4, 10, 12, 13, 2, 7, 14, 6, 5, 9, 0, 11, 15, 8, 1}, {10, 7, 12, 9, 14, 3, 13, 15, 4, 0, 11, 2, 5, 8, 1, 6},
      {12, 13, 9, 11, 15, 10, 14, 8, 7, 2, 5, 3, 0, 1, 6, 4}, {9, 14, 11, 5, 8, 12, 15, 1, 13, 3, 0, 10, 2, 6, 4, 7},
      {11, 15, 5, 0, 1, 9, 8, 6, 14, 10, 2, 12, 3, 4, 7, 13}};
  uint32_t s[16];
  for (int i = 0; i < 8; i++) s[i] = cv[i];
  for (int i = 0; i < 4; i++) s[8 + i] = IV[i];
  s[12] = (uint32_t)counter; s[13] = (uint32_t)(counter >> 32); s[14] = blen; s[15] = flags;
  auto g = [&](int a, int b, int c, int d, uint32_t mx, uint32_t my) {
    s[a] = s[a] + s[b] + mx; s[d] = ror32(s[d] ^ s[a], 16);
    s[c] = s[c] + s[d];      s[b] = ror32(s[b] ^ s[c], 12);
    s[a] = s[a] + s[b] + my; s[d] = ror32(s[d] ^ s[a], 8);
    s[c] = s[c] + s[d];      s[b] = ror32(s[b] ^ s[c], 7);
  };
  for (int r = 0; r < 7; r++) {
    const uint8_t* m = SCHED[r];
    g(0, 4, 8, 12, blk[m[0]], blk[m[1]]);   g(1, 5, 9, 13, blk[m[2]], blk[m[3]]);
    g(2, 6, 10, 14, blk[m[4]], blk[m[5]]);  g(3, 7, 11, 15, blk[m[6]], blk[m[7]]);
    g(0, 5, 10, 15, blk[m[8]], blk[m[9]]);  g(1, 6, 11, 12, blk[m[10]], blk[m[11]]);
    g(2, 7, 8, 13, blk[m[12]], blk[m[13]]); g(3, 4, 9, 14, blk[m[14]], blk[m[15]]);
  }
  for (int i = 0; i < 8; i++) cv[i] = s[i] ^ s[i + 8];
}
static void chunk_cv(const uint8_t* p, size_t len, uint64_t counter, bool root, uint32_t cv[8]) {
  memcpy(cv, IV, 32);
  const size_t nb = len == 0 ? 1 : (len + 63) / 64;
  for (size_t b = 0; b < nb; b++) {
    uint32_t blk[16] = {0};
    const size_t bl = (len - 64 * b) < 64 ? (len - 64 * b) : 64;
    memcpy(blk, p + 64 * b, bl);
    uint32_t flags = (b == 0 ? 1u : 0u) | (b == nb - 1 ? (2u | (root ? 8u : 0u)) : 0u);
    compress(cv, blk, counter, (uint32_t)bl, flags);
  }
}
static void subtree(const uint8_t* p, size_t len, uint64_t chunk0, bool root, uint32_t cv[8]) {
  const size_t nch = len == 0 ? 1 : (len + 1023) / 1024;
  if (nch == 1) { chunk_cv(p, len, chunk0, root, cv); return; }
  size_t left = 1;
  while (left * 2 < nch) left *= 2;            // largest power of two < nch
  uint32_t blk[16];
  subtree(p, 1024 * left, chunk0, false, blk);
  subtree(p + 1024 * left, len - 1024 * left, chunk0 + left, false, blk + 8);
  memcpy(cv, IV, 32);
  compress(cv, blk, 0, 64, 4u | (root ? 8u : 0u));
}
void blake3_host(const uint8_t* in, size_t len, uint8_t out[32]) {
  uint32_t cv[8];
  subtree(in, len, 0, true, cv);
  memcpy(out, cv, 32);
}

}  // namespace lcpc
