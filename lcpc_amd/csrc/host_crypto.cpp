// lcpc_amd/csrc/host_crypto.cpp -- see host_crypto.h
#include "host_crypto.h"
#include <string.h>
#if defined(__x86_64__)
#include <immintrin.h>
#endif

namespace lcpc {

static inline uint64_t rol64(uint64_t x, int n) { return n ? (x << n) | (x >> (64 - n)) : x; }
static inline uint32_t ror32(uint32_t x, int n) { return (x >> n) | (x << (32 - n)); }

static void keccak_f1600_scalar(uint64_t a[25]) {
  static const uint64_t RC[24] = {
      0x0000000000000001ull, 0x0000000000008082ull, 0x800000000000808Aull, 0x8000000080008000ull, 0x000000000000808Bull,
      0x0000000080000001ull, 0x8000000080008081ull, 0x8000000000008009ull, 0x000000000000008Aull, 0x0000000000000088ull,
      0x0000000080008009ull, 0x000000008000000Aull, 0x000000008000808Bull, 0x800000000000008Bull, 0x8000000000008089ull,
      0x8000000000008003ull, 0x8000000000008002ull, 0x8000000000000080ull, 0x000000000000800Aull, 0x800000008000000Aull,
      0x8000000080008081ull, 0x8000000000008080ull, 0x0000000080000001ull, 0x8000000080008008ull};
  // lane (x, y) lives at a[x + 5y]; rho offsets and pi destinations written out so the compiler keeps all 25
  // lanes in registers
  uint64_t a00 = a[0], a10 = a[1], a20 = a[2], a30 = a[3], a40 = a[4], a01 = a[5], a11 = a[6], a21 = a[7], a31 = a[8], a41 = a[9],
           a02 = a[10], a12 = a[11], a22 = a[12], a32 = a[13], a42 = a[14], a03 = a[15], a13 = a[16], a23 = a[17], a33 = a[18],
           a43 = a[19], a04 = a[20], a14 = a[21], a24 = a[22], a34 = a[23], a44 = a[24];
  for (int round = 0; round < 24; round++) {
    const uint64_t c0 = a00 ^ a01 ^ a02 ^ a03 ^ a04, c1 = a10 ^ a11 ^ a12 ^ a13 ^ a14, c2 = a20 ^ a21 ^ a22 ^ a23 ^ a24,
                   c3 = a30 ^ a31 ^ a32 ^ a33 ^ a34, c4 = a40 ^ a41 ^ a42 ^ a43 ^ a44;
    const uint64_t d0 = c4 ^ rol64(c1, 1), d1 = c0 ^ rol64(c2, 1), d2 = c1 ^ rol64(c3, 1), d3 = c2 ^ rol64(c4, 1), d4 = c3 ^ rol64(c0, 1);
    // theta + rho + pi: b[y][2x+3y] = rol(a[x][y] ^ d[x], rho[x][y])
    const uint64_t b00 = a00 ^ d0, b13 = rol64(a01 ^ d0, 36), b21 = rol64(a02 ^ d0, 3), b34 = rol64(a03 ^ d0, 41), b42 = rol64(a04 ^ d0, 18);
    const uint64_t b02 = rol64(a10 ^ d1, 1), b10 = rol64(a11 ^ d1, 44), b23 = rol64(a12 ^ d1, 10), b31 = rol64(a13 ^ d1, 45), b44 = rol64(a14 ^ d1, 2);
    const uint64_t b04 = rol64(a20 ^ d2, 62), b12 = rol64(a21 ^ d2, 6), b20 = rol64(a22 ^ d2, 43), b33 = rol64(a23 ^ d2, 15), b41 = rol64(a24 ^ d2, 61);
    const uint64_t b01 = rol64(a30 ^ d3, 28), b14 = rol64(a31 ^ d3, 55), b22 = rol64(a32 ^ d3, 25), b30 = rol64(a33 ^ d3, 21), b43 = rol64(a34 ^ d3, 56);
    const uint64_t b03 = rol64(a40 ^ d4, 27), b11 = rol64(a41 ^ d4, 20), b24 = rol64(a42 ^ d4, 39), b32 = rol64(a43 ^ d4, 8), b40 = rol64(a44 ^ d4, 14);
    // chi (bXY = lane x of row y), iota
    a00 = b00 ^ (~b10 & b20) ^ RC[round]; a10 = b10 ^ (~b20 & b30); a20 = b20 ^ (~b30 & b40); a30 = b30 ^ (~b40 & b00); a40 = b40 ^ (~b00 & b10);
    a01 = b01 ^ (~b11 & b21); a11 = b11 ^ (~b21 & b31); a21 = b21 ^ (~b31 & b41); a31 = b31 ^ (~b41 & b01); a41 = b41 ^ (~b01 & b11);
    a02 = b02 ^ (~b12 & b22); a12 = b12 ^ (~b22 & b32); a22 = b22 ^ (~b32 & b42); a32 = b32 ^ (~b42 & b02); a42 = b42 ^ (~b02 & b12);
    a03 = b03 ^ (~b13 & b23); a13 = b13 ^ (~b23 & b33); a23 = b23 ^ (~b33 & b43); a33 = b33 ^ (~b43 & b03); a43 = b43 ^ (~b03 & b13);
    a04 = b04 ^ (~b14 & b24); a14 = b14 ^ (~b24 & b34); a24 = b24 ^ (~b34 & b44); a34 = b34 ^ (~b44 & b04); a44 = b44 ^ (~b04 & b14);
  }
  a[0] = a00; a[1] = a10; a[2] = a20; a[3] = a30; a[4] = a40; a[5] = a01; a[6] = a11; a[7] = a21; a[8] = a31; a[9] = a41;
  a[10] = a02; a[11] = a12; a[12] = a22; a[13] = a32; a[14] = a42; a[15] = a03; a[16] = a13; a[17] = a23; a[18] = a33; a[19] = a43;
  a[20] = a04; a[21] = a14; a[22] = a24; a[23] = a34; a[24] = a44;
}


#if defined(__x86_64__)
// Keccak-f[1600] on AVX-512VL: one 64-bit lane per XMM register (32 registers: the 25 lanes, 5 column parities and two
// temporaries fit without spilling, which the 16 general-purpose registers of the scalar form cannot offer), 3-input
// logic (vpternlogq: theta's XOR3, chi's a ^ (~b & c) in one instruction) and native rotates (vprolq).  105-110 instructions
// per round, no shuffles: pi is bookkeeping done by the generator (gen/gen_keccak_x25.py -> keccak_x25_gen.h, made by the Makefile), which also
// allocates the registers and emits the whole permutation as one asm statement (left to the compiler, the intrinsic
// form is spilled 4-9 times per round).  The serial STROBE absorb of prove / verify (lcpc-2d/src/lib.rs:1045-1047) runs
// at the speed of this permutation; timings in DESIGN.md section 6a.
#define LCPC_AVX512VL __attribute__((target("avx512f,avx512vl")))
#include "keccak_x25_gen.h"
// which mix: measured per vendor (see the generator); AMD parts take the xor-heavy one
// (LCPC_KECCAK = portable | tern | xor overrides the choice: tests/test_abi.py runs all three)
static const char* keccak_override() {
  static const char* v = getenv("LCPC_KECCAK");
  return v;
}
static bool cpu_prefers_xor_mix() {
  static const bool xor_mix = keccak_override() ? strcmp(keccak_override(), "xor") == 0 : (bool)__builtin_cpu_is("amd");
  return xor_mix;
}
LCPC_AVX512VL static inline void keccak_x25_permute(uint64_t* st) {
  if (cpu_prefers_xor_mix()) keccak_x25_permute_xor(st); else keccak_x25_permute_tern(st);
}
LCPC_AVX512VL static inline void keccak_x25_absorb(uint64_t* st, uint8_t* blk) {      // st ^= blk; permute; blk = 0
  if (cpu_prefers_xor_mix()) keccak_x25_absorb_xor(st, blk); else keccak_x25_absorb_tern(st, blk);
}
LCPC_AVX512VL static void keccak_f1600_avx512(uint64_t a[25]) { keccak_x25_permute(a); }
static bool cpu_has_avx512() {
  static const bool ok = __builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512vl") &&
                         !(keccak_override() && strcmp(keccak_override(), "portable") == 0);
  return ok;
}
void keccak_f1600(uint64_t a[25]) {
  if (cpu_has_avx512()) keccak_f1600_avx512(a);
  else keccak_f1600_scalar(a);
}
#else
void keccak_f1600(uint64_t a[25]) { keccak_f1600_scalar(a); }
#endif
void keccak_f1600_portable(uint64_t a[25]) { keccak_f1600_scalar(a); }

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
  // prove/verify absorb ~50 bytes per coefficient, n_per_row times in a row (lcpc-2d lib.rs:1045-1047): XOR in
  // runs up to the rate boundary instead of byte-at-a-time
  while (n > 0) {
    size_t take = (size_t)(R - pos_);
    if (take > n) take = n;
    uint8_t* dst = st_.b + pos_;
    for (size_t i = 0; i < take; i++) dst[i] ^= d[i];
    pos_ = (uint8_t)(pos_ + take);
    d += take;
    n -= take;
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
  // prove / verify call this n_per_row times in a row with a 6-byte label and one field element (lib.rs:1045-1047).
  // When the whole operation -- meta-AD header, label, length, AD header, message -- stays inside the current rate
  // block, no permutation (and no pos_begin reset) can happen in between, so the bytes are known up front: one XOR run.
  const size_t total = 2 + llen + 4 + 2 + mlen;
  if (total <= 96 && (size_t)pos_ + total < (size_t)R) {
    uint8_t buf[96];
    size_t o = 0;
    buf[o++] = pos_begin_; buf[o++] = SF_M | SF_A;             // begin_op(meta-AD)
    const uint8_t pb1 = (uint8_t)(pos_ + 1);
    memcpy(buf + o, label, llen); o += llen;
    memcpy(buf + o, le, 4); o += 4;                            // meta_ad(len, more = true): no header
    buf[o++] = pb1; buf[o++] = SF_A;                           // begin_op(AD)
    pos_begin_ = (uint8_t)(pos_ + 2 + llen + 4 + 1);
    cur_flags_ = SF_A;
    memcpy(buf + o, msg, mlen); o += mlen;
    uint8_t* dst = st_.b + pos_;
    for (size_t i = 0; i < o; i++) dst[i] ^= buf[i];
    pos_ = (uint8_t)(pos_ + o);
    return;
  }
  meta_ad(label, llen, false);
  meta_ad(le, 4, true);
  ad(msg, mlen, false);
}
#if defined(__x86_64__)
// The coefficient absorbs of prove / verify (lib.rs:1045-1047, 1066-1068): n_per_row operations of ~46 bytes each, one
// permutation every 3.6 of them, strictly serial.  The bytes of the current rate block are laid down in a staging buffer (plain stores -- a position is written once
// between two permutations, so "store" equals STROBE's XOR) and folded into the state when the block is full.
namespace {
struct Sponge25 {             // state lanes + the staging copy of the current rate block
  alignas(64) uint64_t st[25];
  alignas(64) uint8_t blk[256];
  unsigned pos, pb;
};
constexpr unsigned STROBE_R = 166;
inline void sp_fold(Sponge25& s) {                          // lanes 0..20 cover the rate and the two padding bytes
  for (int i = 0; i < 21; i++) { uint64_t w; memcpy(&w, s.blk + 8 * i, 8); s.st[i] ^= w; }
}
LCPC_AVX512VL inline void sp_flush(Sponge25& s) {           // Transcript::run_f
  s.blk[s.pos] ^= (uint8_t)s.pb;
  s.blk[s.pos + 1] ^= 0x04;
  s.blk[STROBE_R + 1] ^= 0x80;
  keccak_x25_absorb(s.st, s.blk);            // folds lanes 0..20 of the block into the state on the way in, zeroes the block
  s.pos = 0;
  s.pb = 0;
}
LCPC_AVX512VL inline void sp_emit(Sponge25& s, uint8_t b) {
  s.blk[s.pos++] = b;
  if (s.pos == STROBE_R) sp_flush(s);
}
LCPC_AVX512VL inline void sp_emit_bulk(Sponge25& s, const uint8_t* p, size_t len) {     // == sp_emit byte by byte
  while (len) {
    const size_t room = STROBE_R - s.pos, take = len < room ? len : room;
    memcpy(s.blk + s.pos, p, take);
    s.pos += (unsigned)take;
    p += take;
    len -= take;
    if (s.pos == STROBE_R) sp_flush(s);
  }
}
LCPC_AVX512VL inline void sp_begin_op(Sponge25& s, uint8_t flags) {
  const uint8_t h0 = (uint8_t)s.pb;
  s.pb = s.pos + 1;
  sp_emit(s, h0);
  sp_emit(s, flags);
}
}  // namespace
LCPC_AVX512VL static void append_messages_avx512(uint64_t st[25], uint8_t& pos_io, uint8_t& pb_io, const uint8_t* label, size_t llen,
                                               const uint8_t* msgs, size_t mlen, size_t n) {
  Sponge25 s;
  memcpy(s.st, st, 200);
  memset(s.blk, 0, sizeof s.blk);
  s.pos = pos_io;
  s.pb = pb_io;
  const uint32_t l32 = (uint32_t)mlen;
  const uint8_t le[4] = {(uint8_t)l32, (uint8_t)(l32 >> 8), (uint8_t)(l32 >> 16), (uint8_t)(l32 >> 24)};
  const size_t hdr_len = 2 + llen + 4 + 2, total = hdr_len + mlen;
  uint8_t tmpl[32] = {0};                    // [pos_begin, M|A, label, len_le32, pos_begin', A] with the two positions patched per message
  if (hdr_len <= 32) {
    tmpl[1] = 0x10 | 0x02;
    memcpy(tmpl + 2, label, llen);
    memcpy(tmpl + 2 + llen, le, 4);
    tmpl[2 + llen + 5] = 0x02;
  }
  for (size_t i = 0; i < n; i++) {
    const uint8_t* msg = msgs + i * mlen;
    if (s.pos + total < STROBE_R) {          // the whole operation stays inside the block: no permutation, no pos_begin reset
      uint8_t* w = s.blk + s.pos;
      if (hdr_len <= 32) {
        memcpy(w, tmpl, 32);                                       // fixed-size copy of the framing; its zero tail is overwritten below
      } else {
        w[1] = 0x10 | 0x02;
        memcpy(w + 2, label, llen);
        memcpy(w + 2 + llen, le, 4);
        w[2 + llen + 5] = 0x02;
      }
      w[0] = (uint8_t)s.pb;                                        // begin_op(meta-AD, flags M | A): previous pos_begin
      w[2 + llen + 4] = (uint8_t)(s.pos + 1);                      // begin_op(AD): pos_begin of the meta-AD operation
      if (mlen == 32) memcpy(w + hdr_len, msg, 32);                // (the coefficient absorbs of Ft255: inlined moves)
      else memcpy(w + hdr_len, msg, mlen);
      s.pb = s.pos + 2 + (unsigned)llen + 4 + 1;
      s.pos += (unsigned)total;
      continue;
    }
    sp_begin_op(s, 0x10 | 0x02);             // the operation crosses the end of the block: one permutation on the way
    sp_emit_bulk(s, label, llen);
    sp_emit_bulk(s, le, 4);
    sp_begin_op(s, 0x02);
    sp_emit_bulk(s, msg, mlen);
  }
  sp_fold(s);                                // what the open block holds goes into the state, as STROBE keeps it
  memcpy(st, s.st, 200);
  pos_io = (uint8_t)s.pos;
  pb_io = (uint8_t)s.pb;
}
#endif

void Transcript::append_messages(const uint8_t* label, size_t llen, const uint8_t* msgs, size_t mlen, size_t n) {
#if defined(__x86_64__)
  if (cpu_has_avx512() && 2 + llen + 4 + 2 + mlen <= 160) {
    append_messages_avx512(st_.w, pos_, pos_begin_, label, llen, msgs, mlen, n);
    cur_flags_ = SF_A;
    return;
  }
#endif
  for (size_t i = 0; i < n; i++) append_message(label, llen, msgs + i * mlen, mlen);
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
// W blocks side by side (structure of arrays): the quarter-round loops run over the lane index, which the compiler turns
// into vector instructions (AVX-512 / AVX2 / SSE2 clones picked at load time)
#if defined(__x86_64__)
__attribute__((target_clones("avx512f", "avx2", "default")))
#endif
static void chacha20_blocks16(const uint32_t key[8], uint64_t stream, uint64_t block0, uint32_t* out) {
  constexpr int W = 16;
  uint32_t x[16][W], s[16][W];
  for (int l = 0; l < W; l++) {
    const uint64_t ctr = block0 + (uint64_t)l;
    s[0][l] = 0x61707865u; s[1][l] = 0x3320646Eu; s[2][l] = 0x79622D32u; s[3][l] = 0x6B206574u;
    for (int i = 0; i < 8; i++) s[4 + i][l] = key[i];
    s[12][l] = (uint32_t)ctr; s[13][l] = (uint32_t)(ctr >> 32);
    s[14][l] = (uint32_t)stream; s[15][l] = (uint32_t)(stream >> 32);
  }
  for (int i = 0; i < 16; i++)
    for (int l = 0; l < W; l++) x[i][l] = s[i][l];
#define LCPC_QR(a, b, c, d)                                                                     \
  for (int l = 0; l < W; l++) { x[a][l] += x[b][l]; uint32_t t = x[d][l] ^ x[a][l]; x[d][l] = (t << 16) | (t >> 16); } \
  for (int l = 0; l < W; l++) { x[c][l] += x[d][l]; uint32_t t = x[b][l] ^ x[c][l]; x[b][l] = (t << 12) | (t >> 20); } \
  for (int l = 0; l < W; l++) { x[a][l] += x[b][l]; uint32_t t = x[d][l] ^ x[a][l]; x[d][l] = (t << 8) | (t >> 24); }  \
  for (int l = 0; l < W; l++) { x[c][l] += x[d][l]; uint32_t t = x[b][l] ^ x[c][l]; x[b][l] = (t << 7) | (t >> 25); }
  for (int r = 0; r < 10; r++) {
    LCPC_QR(0, 4, 8, 12) LCPC_QR(1, 5, 9, 13) LCPC_QR(2, 6, 10, 14) LCPC_QR(3, 7, 11, 15)
    LCPC_QR(0, 5, 10, 15) LCPC_QR(1, 6, 11, 12) LCPC_QR(2, 7, 8, 13) LCPC_QR(3, 4, 9, 14)
  }
#undef LCPC_QR
  for (int l = 0; l < W; l++)
    for (int i = 0; i < 16; i++) out[16 * l + i] = x[i][l] + s[i][l];
}
static void chacha20_block1(const uint32_t key[8], uint64_t stream, uint64_t ctr, uint32_t* out) {
  uint32_t s[16] = {0x61707865u, 0x3320646Eu, 0x79622D32u, 0x6B206574u};
  for (int i = 0; i < 8; i++) s[4 + i] = key[i];
  s[12] = (uint32_t)ctr; s[13] = (uint32_t)(ctr >> 32);
  s[14] = (uint32_t)stream; s[15] = (uint32_t)(stream >> 32);
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
  for (int i = 0; i < 16; i++) out[i] = x[i] + s[i];
}
void chacha20_keystream(const uint32_t key[8], uint64_t stream, uint64_t block0, uint64_t nblocks, uint32_t* out) {
  uint64_t b = 0;
  for (; b + 16 <= nblocks; b += 16) chacha20_blocks16(key, stream, block0 + b, out + 16 * b);
  for (; b < nblocks; b++) chacha20_block1(key, stream, block0 + b, out + 16 * b);
}
void ChaCha20Rng::refill() {
  chacha20_keystream(key_, stream_, counter_, 4, buf_);
  counter_ += 4;
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
