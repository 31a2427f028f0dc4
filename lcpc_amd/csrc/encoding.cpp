// lcpc_amd/csrc/encoding.cpp -- see encoding.h
#include "encoding.h"
#include <algorithm>
#include <cmath>
#include <mutex>
#include <thread>
#include "host_crypto.h"

namespace lcpc {

// ---- field descriptors (lcpc-test-fields/src/lib.rs:13-59) -----------------------------------------
static FieldDesc g_fields[4];
static std::once_flag g_fields_once;
static void init_field(FieldDesc& f, int id, int L, const uint64_t* p, uint64_t gen) {
  memset(&f, 0, sizeof f);
  f.id = id; f.L = L; f.gen = gen;
  for (int i = 0; i < L; i++) f.p[i] = p[i];
  uint64_t x = 1;
  for (int i = 0; i < 6; i++) x *= 2 - f.p[0] * x;       // Newton: p^-1 mod 2^64
  f.inv = 0 - x;
  uint64_t t[MAXL] = {1, 0, 0, 0};
  for (int rep = 0; rep < 2; rep++) {                     // R, R^2 by doubling
    for (int i = 0; i < 64 * L; i++) h_add(f, t, t, t);
    memcpy(rep == 0 ? f.r : f.r2, t, 8 * L);
  }
  unsigned nb = 64 * (L - 1);
  for (uint64_t v = f.p[L - 1]; v; v >>= 1) nb++;
  f.num_bits = nb;
  f.top_mask = (~(uint64_t)0) >> (64 * L - nb);
  uint64_t e[MAXL];
  memcpy(e, f.p, 8 * L);
  e[0] -= 1;
  unsigned S = 0;
  while ((e[0] & 1) == 0) {
    for (int i = 0; i < L; i++) e[i] = (e[i] >> 1) | (i + 1 < L ? e[i + 1] << 63 : 0);
    S++;
  }
  f.S = S;
  uint64_t g[MAXL] = {gen, 0, 0, 0}, acc[MAXL];
  h_mul(f, g, g, f.r2);
  memcpy(acc, f.r, 8 * L);
  for (int bit = 64 * L - 1; bit >= 0; bit--) {           // ROOT_OF_UNITY = gen^((p-1)/2^S)
    h_mul(f, acc, acc, acc);
    if ((e[bit / 64] >> (bit % 64)) & 1) h_mul(f, acc, acc, g);
  }
  memcpy(f.rou, acc, 8 * L);
}
const FieldDesc* field_desc(int id) {
  std::call_once(g_fields_once, [] {
    const uint64_t p63[1] = {0x46d0760000000001ull};
    const uint64_t p127[2] = {0x7f2bd90000000001ull, 0x6e754097ba20e0bfull};
    const uint64_t p191[3] = {0xd246820000000001ull, 0x936888270ceecbcdull, 0x453708aa3fbc8ddaull};
    const uint64_t p255[4] = {0x02a4f20000000001ull, 0xef73c79086595f30ull, 0xfda9df04b9575969ull, 0x663c799b6e4d2900ull};
    init_field(g_fields[0], 0, 1, p63, 10);
    init_field(g_fields[1], 1, 2, p127, 3);
    init_field(g_fields[2], 2, 3, p191, 5);
    init_field(g_fields[3], 3, 4, p255, 5);
  });
  return (id >= 0 && id < 4) ? &g_fields[id] : nullptr;
}

uint64_t log2_ceil(uint64_t v) { uint64_t l = 0; while (l < 63 && ((uint64_t)1 << l) < v) l++; return l; }
uint64_t next_pow2(uint64_t v) { return (uint64_t)1 << log2_ceil(v); }
uint64_t n_degree_tests(uint64_t lambda, uint64_t len, uint64_t flog2) {
  const uint64_t den = flog2 - log2_ceil(len);
  return (lambda + den - 1) / den;
}

// ---- Ligero ------------------------------------------------------------------------------------------
uint64_t ligero_n_col_opens(uint32_t rn, uint32_t rd) {
  const double rho = (double)rn / (double)rd;
  const double den = std::log2((1.0 + rho) / 2.0);
  return (uint64_t)std::ceil(-128.0 / den);
}
int ligero_get_dims(const FieldDesc& f, uint64_t len, uint32_t rn, uint32_t rd, uint64_t* nr, uint64_t* np, uint64_t* nc) {
  const double rho = (double)rn / (double)rd;
  const uint64_t flog2 = f.flog2();
  const uint64_t n_col_opens = ligero_n_col_opens(rn, rd);
  const double lncf = (double)(n_col_opens * len);
  const double ndt = (double)n_degree_tests(128, (uint64_t)std::ceil(std::sqrt(lncf) / rho), flog2);
  const uint64_t nc1 = next_pow2((uint64_t)std::ceil(std::sqrt(lncf / ndt) / rho));
  if (nc1 > ((uint64_t)1 << f.S)) return -1;
  const uint64_t np1 = nc1 * rn / rd;
  if (np1 == 0) return -1;
  const uint64_t nr1 = (len + np1 - 1) / np1, nd1 = n_degree_tests(128, nc1, flog2);
  const uint64_t nc2 = nc1 / 2, np2 = np1 / 2;
  if (np2 == 0) return -1;
  const uint64_t nr2 = (len + np2 - 1) / np2, nd2 = n_degree_tests(128, nc2, flog2);
  const uint64_t sz1 = n_col_opens * nr1 + (1 + nd1) * np1, sz2 = n_col_opens * nr2 + (1 + nd2) * np2;
  if (sz1 < sz2) { *nr = nr1; *np = np1; *nc = nc1; } else { *nr = nr2; *np = np2; *nc = nc2; }
  return 0;
}
// ---- Brakedown / SDIG ----------------------------------------------------------------------------------
static double ent(double z) { return -z * std::log2(z) - (1.0 - z) * std::log2(1.0 - z); }
bool sdig_spec(int code, SdigSpec* s) {
  static const uint64_t T[6][7] = {{239, 2000, 71, 2500, 71, 50, 20},  {69, 500, 111, 2500, 147, 100, 20},
                                   {89, 500, 61, 1000, 1521, 1000, 20}, {1, 5, 41, 500, 41, 25, 20},
                                   {211, 1000, 97, 1000, 202, 125, 20}, {119, 500, 241, 2000, 43, 25, 20}};
  if (code < 1 || code > 6) return false;
  const uint64_t* t = T[code - 1];
  s->an = t[0]; s->ad = t[1]; s->bn = t[2]; s->bd = t[3]; s->rn = t[4]; s->rd = t[5]; s->baselen = t[6];
  s->alpha = (double)s->an / (double)s->ad;
  s->beta = (double)s->bn / (double)s->bd;
  s->r = (double)s->rn / (double)s->rd;
  s->dist = (double)(s->bn * s->rd) / (double)(s->bd * s->rn);
  s->mu = s->r - 1.0 - s->r * s->alpha;
  s->nu = s->beta + s->alpha * s->beta + 0.03;
  s->cn1 = ent(s->beta) + s->alpha * ent(1.28 * s->beta / s->alpha);
  s->cn2 = s->beta * std::log2(s->alpha / (1.28 * s->beta));
  s->dn1 = s->r * s->alpha * ent(s->beta / s->r) + s->mu * ent(s->nu / s->mu);
  s->dn2 = s->alpha * s->beta * std::log2(s->mu / s->nu);
  return true;
}
uint64_t sdig_n_col_opens(int code) {
  SdigSpec s;
  if (!sdig_spec(code, &s)) return 0;
  return (uint64_t)std::ceil(-128.0 / std::log2(1.0 - s.dist / 3.0));
}
static uint64_t cmd(uint64_t n, uint64_t num, uint64_t den) { return (n * num + den - 1) / den; }
bool sdig_level_dims(const SdigSpec& s, uint64_t n, double log2p, std::vector<LevelDims>& pre, std::vector<LevelDims>& post) {
  pre.clear();
  post.clear();
  if (n <= s.baselen) return false;
  std::vector<uint64_t> tmp;
  for (uint64_t ni = n; ni > s.baselen; ni = cmd(ni, s.an, s.ad)) tmp.push_back(ni);
  tmp.push_back(cmd(tmp.back(), s.an, s.ad));
  for (size_t i = 0; i + 1 < tmp.size(); i++) {
    const uint64_t ni = tmp[i], mi = tmp[i + 1];
    uint64_t cn = std::min(std::max(cmd(ni, 32 * s.bn, 25 * s.bd), 4 + cmd(ni, s.bn, s.bd)),
                           (uint64_t)std::ceil((110.0 / (double)ni + s.cn1) / s.cn2));
    cn = std::min(cn, mi);
    pre.push_back({ni, mi, cn});
    const uint64_t nip = cmd(mi, s.rn, s.rd);
    const uint64_t mip = cmd(ni, s.rn, s.rd) - ni - nip;
    const uint64_t tmp1 = cmd(ni, 2 * s.bn, s.bd);
    const uint64_t tmp2 = cmd(ni, s.rn, s.rd) - ni + 110;
    uint64_t dn = std::min(tmp1 + (uint64_t)std::ceil((double)tmp2 / log2p),
                           (uint64_t)std::ceil((110.0 / (double)ni + s.dn1) / s.dn2));
    dn = std::min(dn, mip);
    post.push_back({nip, mip, dn});
  }
  return true;
}
uint64_t sdig_codeword_length(const std::vector<LevelDims>& pre, const std::vector<LevelDims>& post) {
  uint64_t c = pre[0].n + post.back().n;
  for (size_t i = 0; i + 1 < pre.size(); i++) c += pre[i].m;
  for (auto& d : post) c += d.m;
  return c;
}
// new (lib.rs:103-110) and, with ml = true, new_ml (lib.rs:114-123: the first candidate is rounded up to a power of
// two), both through _new_from_np1 (lib.rs:69-87)
bool sdig_n_per_row(const FieldDesc& f, uint64_t len, int code, uint64_t* out, bool ml) {
  const uint64_t flog2 = f.flog2(), n_col_opens = sdig_n_col_opens(code);
  if (!n_col_opens || !len) return false;
  const double lncf = (double)(n_col_opens * len);
  const double ndt = (double)n_degree_tests(128, (uint64_t)std::ceil(std::sqrt(lncf)) * 2, flog2);
  uint64_t np1 = (uint64_t)std::ceil(std::sqrt(lncf / ndt));
  if (ml) { uint64_t q = 1; while (q < np1) q <<= 1; np1 = q; }       // checked_next_power_of_two
  if (np1 > len) np1 = len;
  const uint64_t nr1 = (len + np1 - 1) / np1, nd1 = n_degree_tests(128, np1 * 2, flog2);
  const uint64_t np2 = np1 / 2;
  if (np2 == 0) return false;
  const uint64_t nr2 = (len + np2 - 1) / np2, nd2 = n_degree_tests(128, np2 * 2, flog2);
  const uint64_t sz1 = n_col_opens * nr1 + (1 + nd1) * np1, sz2 = n_col_opens * nr2 + (1 + nd2) * np2;
  *out = sz1 < sz2 ? np1 : np2;
  return true;
}

// gen_code (matgen.rs:114-188) emits the matrix column by column (= per input); we scatter each
// column's (sorted) entries into per-output buckets afterwards to get CSR-by-output.
static void gen_code_csr(const FieldDesc& f, uint64_t n, uint64_t m, uint64_t d, ChaCha20Rng& rng, CsrMatrix& out) {
  const int L = f.L;
  std::vector<uint32_t> ridx(n * d);
  std::vector<uint64_t> vals(n * d * L);
  std::vector<uint64_t> tmp(d);
  for (uint64_t c = 0; c < n; c++) {
    uint64_t got = 0;
    while (got < d) {
      const uint64_t x = rng.uniform(m);
      bool dup = false;
      for (uint64_t i = 0; i < got; i++) dup |= tmp[i] == x;
      if (!dup) tmp[got++] = x;
    }
    std::sort(tmp.begin(), tmp.end());
    for (uint64_t i = 0; i < d; i++) {
      uint64_t* v = &vals[(c * d + i) * L];
      do rng.field_random(f, v); while (h_is_zero(f, v));
      ridx[c * d + i] = (uint32_t)tmp[i];
    }
  }
  out.n_in = n;
  out.n_out = m;
  out.rowptr.assign(m + 1, 0);
  for (uint64_t k = 0; k < n * d; k++) out.rowptr[ridx[k] + 1]++;
  for (uint64_t o = 0; o < m; o++) out.rowptr[o + 1] += out.rowptr[o];
  out.colidx.resize(n * d);
  out.vals.resize(n * d * L);
  std::vector<uint32_t> fill(out.rowptr.begin(), out.rowptr.end() - 1);
  for (uint64_t c = 0; c < n; c++)
    for (uint64_t i = 0; i < d; i++) {
      const uint64_t k = c * d + i;
      const uint32_t slot = fill[ridx[k]]++;
      out.colidx[slot] = (uint32_t)c;
      memcpy(&out.vals[(uint64_t)slot * L], &vals[k * L], 8 * L);
    }
}
bool sdig_generate(const FieldDesc& f, const SdigSpec& s, uint64_t n_per_row, uint64_t seed, std::vector<CsrMatrix>& pre,
                   std::vector<CsrMatrix>& post, std::vector<LevelDims>& pre_dims, std::vector<LevelDims>& post_dims) {
  if (!sdig_level_dims(s, n_per_row, (double)f.flog2(), pre_dims, post_dims)) return false;
  const size_t t = pre_dims.size();
  pre.assign(t, CsrMatrix());
  post.assign(t, CsrMatrix());
  std::vector<std::thread> th;                         // levels are independent streams (matgen.rs:38-49)
  for (size_t i = 0; i < t; i++)
    th.emplace_back([&, i] {
      ChaCha20Rng rng = ChaCha20Rng::seed_from_u64(seed);
      rng.set_stream((uint64_t)i);
      gen_code_csr(f, pre_dims[i].n, pre_dims[i].m, pre_dims[i].d, rng, pre[i]);
      gen_code_csr(f, post_dims[i].n, post_dims[i].m, post_dims[i].d, rng, post[i]);
    });
  for (auto& x : th) x.join();
  return true;
}

}  // namespace lcpc
