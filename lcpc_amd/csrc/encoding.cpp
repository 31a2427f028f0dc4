// lcpc_amd/csrc/encoding.cpp -- see encoding.h
#include "encoding.h"
#include <exception>
#include <algorithm>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <vector>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "host_crypto.h"
#include "host_par.h"

namespace lcpc {

// a container can show 256 hardware threads and be granted 16 CPUs of time; more threads than that only get throttled
unsigned usable_cores() {
  static const unsigned cached = [] {
    unsigned n = std::thread::hardware_concurrency();
    if (n == 0) n = 1;
    long long q = -1, per = 100000;
    if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r")) {                         // cgroup v2: "<quota|max> <period>"
      char qs[32] = {0};
      if (fscanf(f, "%31s %lld", qs, &per) == 2 && strcmp(qs, "max") != 0) q = atoll(qs);
      fclose(f);
    } else if (FILE* f1 = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) {      // cgroup v1
      if (fscanf(f1, "%lld", &q) != 1) q = -1;
      fclose(f1);
      if (FILE* f2 = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) { if (fscanf(f2, "%lld", &per) != 1) per = 100000; fclose(f2); }
    }
    if (q > 0 && per > 0) { const unsigned lim = (unsigned)std::max<long long>(1, q / per); if (lim < n) n = lim; }
    return n;
  }();
  return cached;
}

// ---- the worker pool behind parallel_for (host_par.h) -------------------------------------------------
namespace {
struct ParPool {
  std::mutex mu;
  std::condition_variable cv;
  std::vector<ParJob*> jobs;                    // regions with unclaimed chunks (and exhausted ones until their caller removes them)
  ParJob* pick() {                              // under mu
    for (ParJob* j : jobs)
      if (j->next.load(std::memory_order_relaxed) < j->n_chunks && j->attached.load(std::memory_order_relaxed) < j->max_workers) return j;
    return nullptr;
  }
  static void drain(ParJob* j) {
    for (;;) {
      const uint64_t c = j->next.fetch_add(1);
      if (c >= j->n_chunks) return;
      try { j->run(j->ctx, c); } catch (...) { j->failed.store(true); }
    }
  }
  void worker() {
    std::unique_lock<std::mutex> lk(mu);
    for (;;) {
      ParJob* j;
      while (!(j = pick())) cv.wait(lk);
      j->attached.fetch_add(1);                 // under mu: the caller removes the job under mu before it waits for attached == 0
      lk.unlock();
      drain(j);
      j->attached.fetch_sub(1);                 // the last touch of *j by this thread
      lk.lock();
    }
  }
};
ParPool* par_pool() {
  // created on first use and never destroyed (detached workers: nothing to join at process exit or library unload)
  static ParPool* pool = [] {
    ParPool* p = new ParPool;
    const unsigned n = usable_cores() > 1 ? usable_cores() - 1 : 0;
    for (unsigned i = 0; i < n; i++) {
      try { std::thread([p] { p->worker(); }).detach(); } catch (...) { break; }     // fewer workers: the callers do more themselves
    }
    return p;
  }();
  return pool;
}
}  // namespace

void par_run(ParJob& job) {
  ParPool* p = par_pool();
  {
    std::lock_guard<std::mutex> g(p->mu);
    p->jobs.push_back(&job);
  }
  if (job.max_workers == 1) p->cv.notify_one(); else p->cv.notify_all();
  ParPool::drain(&job);                         // the caller works too, so a busy (or absent) pool only costs parallelism
  {
    std::lock_guard<std::mutex> g(p->mu);
    for (size_t i = 0; i < p->jobs.size(); i++)
      if (p->jobs[i] == &job) { p->jobs.erase(p->jobs.begin() + i); break; }
  }
  // no new worker can attach now; wait for those still running a chunk (about as long as the caller's own last chunk)
  for (unsigned spin = 0; job.attached.load() != 0; spin++) {
    if (spin > 256) std::this_thread::yield();
  }
}


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

// ---- matgen (matgen.rs:28-52, 114-188) ---------------------------------------------------------------------------------
// gen_code draws, per input column, d distinct output indices (Uniform, re-drawn on collision), sorts them, then one
// non-zero field element per entry (Field::random: masked 64 L-bit candidates, rejected while >= p -- 60 % of the
// candidates of Ft255).  One ChaCha20 stream per level feeds precode then postcode, so WHERE a column's draws start
// depends on every rejection before it.  The reference walks that chain serially; level 0 holds 82 % of the entries at
// C3, so a thread per level (matgen.rs:38-49) hardly helps.  Here:
//   1. the keystream of a level is materialised in bulk (seekable cipher: 16 blocks per vector pass, all cores);
//   2. ONE cheap serial pass replays only the accept / reject decisions (one or two compares per candidate) and
//      records the stream position at every chunk of columns;
//   3. the chunks are then generated in parallel from their recorded positions (sort, copy the accepted values), and
//   4. the CSC-by-input entries are transposed to CSR-by-output with a parallel stable counting sort.
// Steps 2 and 3 are the same template (WRITE = false / true), so they cannot disagree about the stream.
namespace {

struct LevelStream {
  uint32_t key[8];
  uint64_t stream = 0;
  uint64_t* w = nullptr;            // w[i] = the i-th next_u64() of the level's RNG
  uint64_t n_valid = 0;
  ~LevelStream() { free(w); }
  void ensure(uint64_t n) {         // at least n values available
    if (n <= n_valid) return;
    uint64_t target = std::max<uint64_t>(n + (n >> 6) + 4096, n_valid);
    target = (target + 127) & ~(uint64_t)127;                               // whole groups of 16 blocks (8 values per block)
    uint64_t* nw = static_cast<uint64_t*>(realloc(w, target * 8));          // (not value-initialised: filled right below)
    if (!nw) throw std::bad_alloc();
    w = nw;
    const uint64_t b0 = n_valid / 8, b1 = target / 8;
    uint32_t* out = reinterpret_cast<uint32_t*>(w);                         // little-endian: value = word 2i | word 2i+1 << 32
    parallel_for(b1 - b0, 4096, [&](uint64_t x, uint64_t y) { chacha20_keystream(key, stream, b0 + x, y - x, out + 16 * (b0 + x)); });
    n_valid = target;
  }
};

// rand 0.8 Uniform::<usize>::new(0, m).sample [3P] on the materialised stream
struct UniformM {
  uint64_t range, zone;
  explicit UniformM(uint64_t m) : range(m) { zone = UINT64_MAX - (UINT64_MAX - m + 1) % m; }
};

// columns [c0, c1) of one matrix from stream position `pos`; returns the position after them.  WRITE = false: decisions
// only.  `need`: called with the index bound before every read burst (extends the stream in the serial pass).
template <bool WRITE, typename Need>
uint64_t gen_columns(const FieldDesc& f, const uint64_t* const& ks, uint64_t pos, uint64_t c0, uint64_t c1, const UniformM& um, uint64_t d,
                     uint32_t* ridx, uint64_t* vals, Need need) {
  const int L = f.L;
  const uint64_t ptop = f.p[L - 1], mask = f.top_mask;
  uint64_t tmp[64];
  for (uint64_t c = c0; c < c1; c++) {
    uint64_t got = 0;
    while (got < d) {                                          // d distinct indices (matgen.rs:119-131)
      need(pos + 1);
      const uint64_t v = ks[pos++];
      const u128 mm = (u128)v * um.range;
      if ((uint64_t)mm > um.zone) continue;
      const uint64_t x = (uint64_t)(mm >> 64);
      bool dup = false;
      for (uint64_t i = 0; i < got; i++) dup |= tmp[i] == x;
      if (!dup) tmp[got++] = x;
    }
    if (WRITE) {
      std::sort(tmp, tmp + d);
      for (uint64_t i = 0; i < d; i++) ridx[c * d + i] = (uint32_t)tmp[i];
    }
    for (uint64_t i = 0; i < d; i++) {                         // one non-zero element per entry (matgen.rs:148-180)
      for (;;) {
        need(pos + L);
        const uint64_t top = ks[pos + L - 1] & mask;
        bool ok = top < ptop;
        if (!ok && top == ptop) {                              // compare the lower limbs (rare)
          uint64_t t[MAXL];
          for (int j = 0; j < L; j++) t[j] = ks[pos + j];
          t[L - 1] = top;
          ok = !h_ge_p(f, t);
        }
        if (ok) {                                              // zero is re-drawn
          bool nz = top != 0;
          for (int j = 0; j + 1 < L && !nz; j++) nz = ks[pos + j] != 0;
          ok = nz;
        }
        if (ok) {
          if (WRITE) {
            uint64_t* v = vals + (c * d + i) * L;
            for (int j = 0; j + 1 < L; j++) v[j] = ks[pos + j];
            v[L - 1] = top;
          }
          pos += L;
          break;
        }
        pos += L;
      }
    }
  }
  return pos;
}

struct MatJob {
  uint64_t n, m, d;
  uint64_t chunk;                     // columns per chunk
  std::vector<uint64_t> start;        // stream position of every chunk
  CsrMatrix* out;
};

// entries (column-major, sorted inside a column) -> CSR by output, columns ascending inside a row: stable counting sort
void transpose_to_csr(const FieldDesc& f, const MatJob& j, const RawBuf<uint32_t>& ridx, const RawBuf<uint64_t>& vals) {
  const int L = f.L;
  const uint64_t n = j.n, m = j.m, d = j.d, nchunks = j.start.size();
  CsrMatrix& out = *j.out;
  out.n_in = n;
  out.n_out = m;
  out.rowptr.alloc(m + 1);
  out.rowptr[0] = 0;
  out.colidx.alloc(n * d);
  out.vals.alloc(n * d * L);
  RawBuf<uint32_t> hist;                                       // hist[chunk][o] -> later: first slot of (chunk, o)
  hist.alloc(nchunks * m);
  parallel_for(nchunks, 1, [&](uint64_t a, uint64_t b) {
    for (uint64_t ch = a; ch < b; ch++) {
      uint32_t* h = &hist[ch * m];
      memset(h, 0, m * 4);
      const uint64_t k1 = std::min(n, (ch + 1) * j.chunk) * d;
      for (uint64_t k = ch * j.chunk * d; k < k1; k++) h[ridx[k]]++;
    }
  });
  parallel_for(m, 2048, [&](uint64_t a, uint64_t b) {          // per output: its total, and each chunk's offset inside the row
    for (uint64_t o = a; o < b; o++) {
      uint32_t run = 0;
      for (uint64_t ch = 0; ch < nchunks; ch++) { const uint32_t t = hist[ch * m + o]; hist[ch * m + o] = run; run += t; }
      out.rowptr[o + 1] = run;
    }
  });
  for (uint64_t o = 0; o < m; o++) out.rowptr[o + 1] += out.rowptr[o];
  parallel_for(nchunks, 1, [&](uint64_t a, uint64_t b) {
    for (uint64_t ch = a; ch < b; ch++) {
      uint32_t* h = &hist[ch * m];
      const uint64_t c1 = std::min(n, (ch + 1) * j.chunk);
      for (uint64_t c = ch * j.chunk; c < c1; c++)
        for (uint64_t i = 0; i < d; i++) {
          const uint64_t k = c * d + i;
          const uint32_t o = ridx[k];
          const uint32_t slot = out.rowptr[o] + h[o]++;
          out.colidx[slot] = (uint32_t)c;
          memcpy(&out.vals[(uint64_t)slot * L], &vals[k * L], 8 * L);
        }
    }
  });
}

// one level: precode then postcode from ONE stream (matgen.rs:43-48)
void gen_level(const FieldDesc& f, uint64_t seed, uint64_t level, const LevelDims& pd, const LevelDims& qd, CsrMatrix& pre, CsrMatrix& post) {
  LevelStream ks;
  {
    ChaCha20Rng rng = ChaCha20Rng::seed_from_u64(seed);
    memcpy(ks.key, rng.key(), 32);
    ks.stream = level;
  }
  MatJob jobs[2] = {{pd.n, pd.m, pd.d, 0, {}, &pre}, {qd.n, qd.m, qd.d, 0, {}, &post}};
  // expected consumption: 1 value per index, L / P(accept) per element; the serial pass extends the stream if it runs short
  double p_acc = 1.0;
  {
    const double ptop = (double)f.p[f.L - 1], range = (double)f.top_mask + 1.0;
    p_acc = ptop / range;
  }
  const double per_entry = 1.02 + (double)f.L / p_acc;
  ks.ensure((uint64_t)((double)(pd.n * pd.d + qd.n * qd.d) * per_entry * 1.01) + 65536);
  auto need = [&](uint64_t bound) { if (bound > ks.n_valid) ks.ensure(bound + 65536); };
  uint64_t pos = 0;
  for (MatJob& j : jobs) {                                     // step 2: positions of the chunks
    const uint64_t target_chunks = 64;
    j.chunk = std::max<uint64_t>(64, (j.n + target_chunks - 1) / target_chunks);
    const UniformM um(j.m);
    for (uint64_t c0 = 0; c0 < j.n; c0 += j.chunk) {
      j.start.push_back(pos);
      pos = gen_columns<false>(f, const_cast<const uint64_t* const&>(ks.w), pos, c0, std::min(j.n, c0 + j.chunk), um, j.d, nullptr, nullptr, need);
    }
  }
  RawBuf<uint32_t> ridx;                                       // one pair of entry buffers serves both matrices (no fresh pages)
  RawBuf<uint64_t> vals;
  ridx.alloc(std::max(jobs[0].n * jobs[0].d, jobs[1].n * jobs[1].d));
  vals.alloc(std::max(jobs[0].n * jobs[0].d, jobs[1].n * jobs[1].d) * f.L);
  for (MatJob& j : jobs) {                                     // steps 3 and 4
    const UniformM um(j.m);
    parallel_for(j.start.size(), 1, [&](uint64_t a, uint64_t b) {
      for (uint64_t ch = a; ch < b; ch++)
        gen_columns<true>(f, const_cast<const uint64_t* const&>(ks.w), j.start[ch], ch * j.chunk, std::min(j.n, (ch + 1) * j.chunk), um, j.d, ridx.data(), vals.data(), [](uint64_t) {});
    });
    transpose_to_csr(f, j, ridx, vals);
  }
}

}  // namespace

bool sdig_generate(const FieldDesc& f, const SdigSpec& s, uint64_t n_per_row, uint64_t seed, std::vector<CsrMatrix>& pre,
                   std::vector<CsrMatrix>& post, std::vector<LevelDims>& pre_dims, std::vector<LevelDims>& post_dims) {
  if (!sdig_level_dims(s, n_per_row, (double)f.flog2(), pre_dims, post_dims)) return false;
  const size_t t = pre_dims.size();
  for (size_t i = 0; i < t; i++)
    if (pre_dims[i].d > 64 || post_dims[i].d > 64) return false;           // (the specs keep d <= 31)
  pre.clear(); post.clear();
  pre.resize(t); post.resize(t);
  // level 0 carries most of the entries: it gets this thread (and, inside its parallel steps, all cores); the remaining
  // levels run beside its serial pass on a second thread (levels are independent streams, matgen.rs:38-49)
  // Either thread may throw (std::bad_alloc from the level buffers): the helper's exception is carried back to this thread,
  // and the helper is joined on every path -- a joinable std::thread must never be destroyed, and an exception escaping a
  // thread body would call std::terminate past lcpc_ctx_create's LCPC_TRY / LCPC_CATCH
  std::exception_ptr rest_ex;
  std::thread rest([&] {
    try {
      for (size_t i = 1; i < t; i++) gen_level(f, seed, i, pre_dims[i], post_dims[i], pre[i], post[i]);
    } catch (...) {
      rest_ex = std::current_exception();
    }
  });
  struct Join { std::thread& th; ~Join() { if (th.joinable()) th.join(); } } join{rest};
  gen_level(f, seed, 0, pre_dims[0], post_dims[0], pre[0], post[0]);
  rest.join();
  if (rest_ex) std::rethrow_exception(rest_ex);
  return true;
}

}  // namespace lcpc
